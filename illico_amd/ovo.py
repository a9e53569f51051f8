"""One-versus-reference dispatchers: HIP replacements of illico/ovo/dense_ovo.py:65-137 and
illico/ovo/sparse_ovo.py:163-210,214-260, with the reference's dispatcher signature
(illico/asymptotic_wilcoxon.py:59-67).  Each returns three float64 [n_groups, chunk_ub-chunk_lb]
planes (pvalues, statistics, fold_change)."""
from __future__ import annotations

from illico_amd._lib import get_engine
from illico_amd.utils.registry import KernelDataFormat, Test, dispatcher_registry


def _run_dense(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, out=None, engine=None):
    eng = engine or get_engine()
    eng.set_groups(grpc)
    return eng.run_dense(X, int(chunk_lb), int(chunk_ub), is_log1p=is_log1p, use_continuity=use_continuity,
                         tie_correct=tie_correct, alternative=alternative, out=out)


def _run_sparse(fmt, X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, out=None, engine=None):
    eng = engine or get_engine()
    eng.set_groups(grpc)
    return eng.run_sparse(fmt, X.data, X.indices, X.indptr, X.shape, int(chunk_lb), int(chunk_ub), is_log1p=is_log1p,
                          use_continuity=use_continuity, tie_correct=tie_correct, alternative=alternative, out=out)


@dispatcher_registry.register(Test.OVO, KernelDataFormat.DENSE)
def dense_ovo_mwu_kernel_over_contiguous_col_chunk(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity=True,
                                                   tie_correct=True, alternative="two-sided", **kw):
    assert grpc.encoded_ref_group != -1
    return _run_dense(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, **kw)


@dispatcher_registry.register(Test.OVO, KernelDataFormat.CSC)
def csc_ovo_mwu_kernel_over_contiguous_col_chunk(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity=True,
                                                 tie_correct=True, alternative="two-sided", **kw):
    assert grpc.encoded_ref_group != -1
    return _run_sparse("csc", X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, **kw)


@dispatcher_registry.register(Test.OVO, KernelDataFormat.CSR)
def csr_ovo_mwu_kernel_over_contiguous_col_chunk(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity=True,
                                                 tie_correct=True, alternative="two-sided", **kw):
    assert grpc.encoded_ref_group != -1
    return _run_sparse("csr", X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative, **kw)
