"""The reference-side ctypes stub of INTEGRATION.md, EXECUTED: the code block is cut out of the document, `illico.utils.registry`
is aliased to a scratch registry with this repository's enums, and the reference-generated goldens run through the six
dispatchers it registers -- dense, CSC, CSR x OVO, OVR -- chunk by chunk from four threads, the way the reference's driver
calls a dispatcher (illico/asymptotic_wilcoxon.py:29-68, 236-241; illico/utils/registry.py:122-139).  A sparse matrix must
reach the device once per matrix, not once per chunk (illico_csr_bind / illico_csc_bind)."""
import ctypes
import re
import sys
import types
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match, load_golden, make_counts, make_labels

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def stub():
    from illico_amd.utils import registry as ours
    text = (ROOT / "INTEGRATION.md").read_text()
    code = re.search(r"```python\n(# illico/hip_backend\.py.*?)```", text, re.S).group(1)
    so = ROOT / "illico_amd" / "csrc" / "libillico_hip.so"
    assert 'ctypes.CDLL("libillico_hip.so")' in code
    code = code.replace('ctypes.CDLL("libillico_hip.so")', f'ctypes.CDLL({str(so)!r})')  # (not on the loader's path here)
    scratch = ours.DispatcherRegistry()
    fake = types.ModuleType("illico.utils.registry")
    fake.KernelDataFormat, fake.Test, fake.dispatcher_registry = ours.KernelDataFormat, ours.Test, scratch
    saved = {k: sys.modules.get(k) for k in ("illico", "illico.utils", "illico.utils.registry")}
    sys.modules["illico"] = types.ModuleType("illico")
    sys.modules["illico.utils"] = types.ModuleType("illico.utils")
    sys.modules["illico.utils.registry"] = fake
    ns = {"__name__": "illico.hip_backend"}
    try:
        exec(compile(code, "INTEGRATION.md:hip_backend", "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    assert len(scratch) == 6
    return types.SimpleNamespace(registry=scratch, ns=ns, Test=ours.Test, Fmt=ours.KernelDataFormat, CSC=ours.CSCMatrix, CSR=ours.CSRMatrix)


def _input_bytes(stub):
    n = ctypes.c_int64(0)
    lib = stub.ns["_lib"]
    lib.illico_profile_input_bytes.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
    assert lib.illico_profile_input_bytes(stub.ns["_ctx"], ctypes.byref(n)) == 0
    return n.value


def _chunks(stub, fmt, test, X, grpc, width, n_threads=4, **kw):
    """What the reference's driver does: one dispatcher call per gene chunk, from threads, every call on the same container."""
    disp = stub.registry.get(test, fmt)
    G, M = grpc.counts.size, X.shape[1] if fmt == stub.Fmt.DENSE else X.shape[1]
    bounds = list(range(0, M, width)) + [M]
    out = [np.empty((G, M)) for _ in range(3)]

    def one(lb, ub):
        p, u, fc = disp(X, lb, ub, grpc, kw.get("is_log1p", False), kw.get("use_continuity", True), kw.get("tie_correct", True),
                        kw.get("alternative", "two-sided"))
        for o, a in zip(out, (p, u, fc)):
            assert a.shape == (G, ub - lb) and a.dtype == np.float64 and a.flags.c_contiguous
            o[:, lb:ub] = a

    with ThreadPoolExecutor(n_threads) as ex:
        list(ex.map(lambda b: one(*b), zip(bounds[:-1], bounds[1:])))
    return tuple(out)


def _container(stub, fmt, X):
    if fmt == stub.Fmt.DENSE:
        return np.ascontiguousarray(X)
    M = sparse.csc_matrix(X) if fmt == stub.Fmt.CSC else sparse.csr_matrix(X)
    M.sort_indices()
    return (stub.CSC if fmt == stub.Fmt.CSC else stub.CSR)(M.data, M.indices, M.indptr, M.shape)


@pytest.mark.parametrize("name", ["c1_1k_200_10", "sparse90"])
@pytest.mark.parametrize("fmt", ["dense", "csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_stub_dispatchers_reproduce_the_reference_goldens(stub, name, fmt, test):
    from illico_amd.utils.groups import encode_and_count_groups
    z = load_golden(name)
    X, labels, ref = z["X"], z["labels"], str(z["reference"])
    F = stub.Fmt(fmt)
    C = _container(stub, F, X)
    _, g = encode_and_count_groups(labels, ref if test == "ovo" else None)
    keys = [k for k in z.files if k.startswith(f"{fmt}|{test}|") and k.count("|") == 4]
    assert keys
    for key in keys:
        _, _, alt, cc, tc = key.split("|")
        got = _chunks(stub, F, stub.Test(test), C, g, 37, use_continuity=bool(int(cc)), tie_correct=bool(int(tc)), alternative=alt)
        gold = z[key]
        assert_planes_match(got, (gold[:, :, 0], gold[:, :, 1], gold[:, :, 2]), ref_row=g.encoded_ref_group, what=f"stub {name} {key}")


@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_stub_moves_a_sparse_matrix_to_the_device_once(stub, fmt):
    """32 chunk calls over ONE host matrix: one upload (the whole CSR matrix used to go up in every call); a second matrix
    replaces the first; a new GroupContainer at whatever id() is picked up (the stub keeps the object, not its id)."""
    X, rng = make_counts(41, 3000, 512, 0.9)
    labels = make_labels(rng, 3000, 7, n_ref=250)
    F = stub.Fmt(fmt)
    C = _container(stub, F, X)
    matrix_bytes = C.data.nbytes + C.indices.nbytes + C.indptr.nbytes
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X, g)
    b0 = _input_bytes(stub)
    got = _chunks(stub, F, stub.Test.OVO, C, g, 16)  # 32 chunks, 4 threads
    assert _input_bytes(stub) - b0 == matrix_bytes
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"stub {fmt} 32 chunks")
    got = _chunks(stub, F, stub.Test.OVO, C, g, 64)  # the same container again: nothing moves
    assert _input_bytes(stub) - b0 == matrix_bytes
    # another matrix, and groups rebuilt until one lands on a recycled id
    X2, _ = make_counts(42, 3000, 512, 0.9)
    C2 = _container(stub, F, X2)
    seen, g2 = {id(g)}, None
    del g
    for k in range(50):
        lab = labels.copy()
        lab[k] = "pert_00001" if lab[k] != "pert_00001" else "pert_00002"
        _, g2 = oracle.encode_and_count_groups(lab, None)
        got2 = _chunks(stub, F, stub.Test.OVR, C2, g2, 128)
        assert_planes_match(got2, oracle.run(X2, g2), what=f"stub {fmt} regrouped {k}")
        if id(g2) in seen:
            break
        seen.add(id(g2))
        g2 = None
    assert _input_bytes(stub) - b0 == matrix_bytes + C2.data.nbytes + C2.indices.nbytes + C2.indptr.nbytes
