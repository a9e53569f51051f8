"""One-versus-rest dispatchers: HIP replacements of illico/ovr/dense_ovr.py:15-80 and
illico/ovr/sparse_ovr.py:100-155,158-208 (same signature as illico_amd/ovo.py)."""
from __future__ import annotations

from illico_amd.ovo import _run_dense, _run_sparse
from illico_amd.utils.registry import KernelDataFormat, Test, dispatcher_registry


@dispatcher_registry.register(Test.OVR, KernelDataFormat.DENSE)
def dense_ovr_mwu_kernel_over_contiguous_col_chunk(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity=True,
                                                   tie_correct=True, alternative="two-sided", **kw):
    assert grpc.encoded_ref_group == -1
    return _run_dense(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, **kw)


@dispatcher_registry.register(Test.OVR, KernelDataFormat.CSC)
def csc_ovr_mwu_kernel_over_contiguous_col_chunk(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity=True,
                                                 tie_correct=True, alternative="two-sided", **kw):
    assert grpc.encoded_ref_group == -1
    return _run_sparse("csc", X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, **kw)


@dispatcher_registry.register(Test.OVR, KernelDataFormat.CSR)
def csr_ovr_mwu_kernel_over_contiguous_col_chunk(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity=True,
                                                 tie_correct=True, alternative="two-sided", **kw):
    assert grpc.encoded_ref_group == -1
    return _run_sparse("csr", X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, **kw)
