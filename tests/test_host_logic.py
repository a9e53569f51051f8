"""CPU-only tests of the host side: C-ABI surface, group encoding, registries, argument errors."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import oracle
from conftest import ROOT, load_golden, make_counts, make_labels


def test_cabi_exports_every_declared_symbol():
    """libillico_hip.so loads without a GPU and exports every entry point include/illico_hip.h declares."""
    from illico_amd import _lib
    header = (ROOT / "include" / "illico_hip.h").read_text()
    declared = sorted(set(re.findall(r"\b(illico_[a-z_0-9]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in illico_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.illico_version().startswith(b"illico_hip")
    assert lib.illico_profile_num_kernels() > 0
    assert lib.illico_profile_kernel_name(0)


def test_no_cpu_fallback_without_gpu():
    """Without a device the product path fails loudly; it never routes through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    X, rng = make_counts(0, 50, 4, 0.5)
    adata = AnnDataLite(X, obs=pd.DataFrame({"pert": make_labels(rng, 50, 3)}))
    with pytest.raises(RuntimeError, match="no CPU fallback|illico_ctx_create"):
        asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert")
    import illico_amd, sys
    src = "".join(p.read_text() for p in (ROOT / "illico_amd").rglob("*.py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_encode_and_count_groups_matches_reference_semantics():
    from illico_amd.utils.groups import encode_and_count_groups
    labels = np.array(["pert_10", "pert_2", "ctrl", "pert_10", "ctrl", "pert_2", "pert_2"])
    uniq, g = encode_and_count_groups(labels, "ctrl")
    assert list(uniq) == ["ctrl", "pert_10", "pert_2"]  # lexicographic, as np.unique (groups.py:42)
    assert g.encoded_ref_group == 0
    np.testing.assert_array_equal(g.encoded_groups, [1, 2, 0, 1, 0, 2, 2])
    np.testing.assert_array_equal(g.counts, [2, 2, 3])
    np.testing.assert_array_equal(g.indptr, [0, 2, 4, 7])
    for k in range(3):
        assert set(g.indices[g.indptr[k]:g.indptr[k + 1]]) == set(np.flatnonzero(g.encoded_groups == k))
    assert all(a.dtype == np.int64 for a in (g.encoded_groups, g.counts, g.indices, g.indptr))
    _, g2 = encode_and_count_groups(labels, None)
    assert g2.encoded_ref_group == -1
    with pytest.raises(ValueError, match="not present"):
        encode_and_count_groups(labels, "nope")
    # same container as the oracle's restatement of groups.py
    z = load_golden("c1_1k_200_10")
    u1, a = encode_and_count_groups(z["labels"], str(z["reference"]))
    u2, b = oracle.encode_and_count_groups(z["labels"], str(z["reference"]))
    np.testing.assert_array_equal(u1, z["groups"])
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_registries_and_handlers():
    from illico_amd.utils.registry import (CSCMatrix, CSRMatrix, KernelDataFormat, Test, data_handler_registry,
                                           dispatcher_registry)
    for t in Test:
        for f in KernelDataFormat:
            assert callable(dispatcher_registry.get(t, f))
            assert callable(dispatcher_registry.get(t.value, f.value))
    with pytest.raises(KeyError, match="No dispatcher registered"):
        dispatcher_registry.pop((Test.OVO, KernelDataFormat.CSR)) and dispatcher_registry.get("ovo", "csr")
    import illico_amd.ovo as ovo
    dispatcher_registry[(Test.OVO, KernelDataFormat.CSR)] = ovo.csr_ovo_mwu_kernel_over_contiguous_col_chunk
    X = np.zeros((4, 3), dtype=np.float32)
    h = data_handler_registry.get(X)
    assert h.kernel_data_format() == KernelDataFormat.DENSE and h.footprint() == X.nbytes
    assert h.fetch(1, 2) == (X, (1, 2)) or h.fetch(1, 2)[1] == (1, 2)
    hc = data_handler_registry.get(sparse.csc_matrix(X))
    assert hc.kernel_data_format() == KernelDataFormat.CSC and isinstance(hc.to_nb(hc.data), CSCMatrix)
    hr = data_handler_registry.get(sparse.csr_matrix(X))
    assert hr.kernel_data_format() == KernelDataFormat.CSR and isinstance(hr.to_nb(hr.data), CSRMatrix)
    with pytest.raises(KeyError, match="is not implemented"):
        data_handler_registry.get(sparse.coo_matrix(X))
    with pytest.raises(KeyError, match="is not implemented"):
        data_handler_registry.get([[1.0]])


def test_sorted_check_vectorised_vs_oracle():
    from illico_amd.utils.ranking import check_indices_sorted_per_parcel
    rng = np.random.RandomState(0)
    for t in range(50):
        M = sparse.random(30, 20, density=0.3, format="csr", random_state=rng)
        if t % 2:
            r = rng.randint(0, 30)
            s, e = M.indptr[r], M.indptr[r + 1]
            M.indices[s:e] = rng.permutation(M.indices[s:e])
        assert check_indices_sorted_per_parcel(M.indices, M.indptr) == oracle.check_indices_sorted_per_parcel(M.indices, M.indptr)
    assert check_indices_sorted_per_parcel(np.array([], dtype=np.int32), np.array([0, 0, 0]))


def test_value_dtype_widening():
    from illico_amd._lib import normalize_values
    assert normalize_values(np.zeros(3, np.uint8)).dtype == np.int32
    assert normalize_values(np.zeros(3, np.int16)).dtype == np.int32
    assert normalize_values(np.zeros(3, np.float16)).dtype == np.float32
    assert normalize_values(np.zeros(3, np.uint32)).dtype == np.int64
    assert normalize_values(np.zeros(3, np.float64)).dtype == np.float64
    with pytest.raises(KeyError):
        normalize_values(np.zeros(3, np.complex64))


def test_argument_errors_before_any_device_work():
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    X, rng = make_counts(0, 60, 300, 0.5)
    labels = make_labels(rng, 60, 3)
    adata = AnnDataLite(X, obs=pd.DataFrame({"pert": labels}))
    with pytest.raises(ValueError, match="not present"):
        asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", reference="nope")
    with pytest.raises(ValueError, match="Invalid batch_size"):
        asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", batch_size="big")
    with pytest.raises(ValueError, match="Unsupported alternative"):
        asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", alternative="bigger")
    with pytest.raises(KeyError, match="is not implemented"):
        asymptotic_wilcoxon(AnnDataLite(sparse.coo_matrix(X), obs=pd.DataFrame({"pert": labels})), is_log1p=False,
                            group_keys="pert")
    bad = sparse.csr_matrix(X)
    s, e = bad.indptr[0], bad.indptr[1]
    bad.indices[s:e] = bad.indices[s:e][::-1].copy()
    with pytest.raises(ValueError, match="not sorted"):
        asymptotic_wilcoxon(AnnDataLite(bad, obs=pd.DataFrame({"pert": labels})), is_log1p=False, group_keys="pert")


def test_shard_bounds():
    from illico_amd.distributed import rank_gene_range, shard_bounds
    for n in (0, 1, 7, 8000, 30000):
        for parts in (1, 2, 3, 8):
            b = shard_bounds(n, parts)
            assert len(b) == parts  # always `parts` windows (empty ones allowed): ranks issue the same number of gathers
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
            w = [u - l for l, u in b]
            assert max(w) - min(w) <= 1
        r = [rank_gene_range(n, k, 8) for k in range(8)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(7))


def test_memmap_handler_is_backed_and_fetches_chunks(tmp_path):
    from illico_amd.utils.registry import KernelDataFormat, data_handler_registry
    X, _ = make_counts(0, 40, 12, 0.5)
    np.save(tmp_path / "x.npy", X)
    mm = np.load(tmp_path / "x.npy", mmap_mode="r")
    h = data_handler_registry.get(mm)
    assert h.streams and h.kernel_data_format() == KernelDataFormat.DENSE
    chunk, bounds = h.fetch(3, 9)   # a lazy column view: its pages are read when the chunk is staged (illico_amd/streaming.py)
    assert bounds == (0, 6) and chunk.shape == (40, 6)
    np.testing.assert_array_equal(np.asarray(chunk), X[:, 3:9])
    assert type(h.to_nb(chunk)) is np.ndarray
    assert not data_handler_registry.get(X).streams


def test_categorical_group_column_encodes_like_strings():
    from illico_amd.utils.groups import encode_and_count_groups
    rng = np.random.RandomState(0)
    labels = np.array([f"pert_{i}" for i in rng.randint(0, 25, size=500)] + ["ctrl"] * 20)
    rng.shuffle(labels)
    # categories in a non-sorted order, with one unused category
    cats = list(np.unique(labels)[::-1]) + ["never_used"]
    col = pd.Series(pd.Categorical(labels, categories=cats))
    for ref in ("ctrl", None, "pert_10"):
        u1, a = encode_and_count_groups(labels, ref)
        u2, b = encode_and_count_groups(col, ref)
        np.testing.assert_array_equal(u1, u2)
        assert u2.dtype.kind == "U"
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    with pytest.raises(ValueError, match="not present"):
        encode_and_count_groups(col, "never_used")
    # a plain object / string Series takes the generic path
    u3, c = encode_and_count_groups(pd.Series(labels), "ctrl")
    np.testing.assert_array_equal(u3, np.unique(labels))


def test_out_of_core_handlers_are_duck_typed():
    """The reference's backed containers (h5py.Dataset, anndata's _CSCDataset; registry.py:162-188) are not in this image: any
    container with the same duck type can be registered under the same handlers (tests/test_gpu_out_of_core.py runs them)."""
    from illico_amd.utils.registry import (H5pyBackedCSCDataHandler, H5pyDatasetDataHandler, KernelDataFormat,
                                           data_handler_registry)

    class Dense:
        shape, dtype = (5, 8), np.dtype(np.float32)

        def __getitem__(self, key):
            return np.arange(40, dtype=np.float32).reshape(5, 8)[key]

    data_handler_registry[Dense] = H5pyDatasetDataHandler
    try:
        h = data_handler_registry.get(Dense())
        assert h.streams and h.kernel_data_format() == KernelDataFormat.DENSE and h.footprint() == 160
        chunk, bounds = h.fetch(2, 5)
        assert bounds == (0, 3) and chunk.shape == (5, 3) and chunk[1, 0] == 10.0
    finally:
        data_handler_registry.pop(Dense)
    assert H5pyBackedCSCDataHandler.streams and H5pyBackedCSCDataHandler(None).kernel_data_format() == KernelDataFormat.CSC


@pytest.mark.parametrize("kind", ["U", "object", "series", "series-object"])
def test_hashed_group_encoding_equals_np_unique(kind):
    """Above 4096 string labels the codes come from one hash pass (pandas.factorize) and a sort of the DISTINCT labels; the container
    must be the one np.unique gives (groups.py:18-58: labels in np.unique's order), missing reference and odd inputs included."""
    from illico_amd.utils import groups as gm
    rng = np.random.RandomState(5)
    n, G = 20000, 300
    codes = rng.randint(0, G, size=n)
    labels = np.where(codes == 0, "non-targeting", np.char.add("pert_", np.char.zfill((codes * 7919 % 1000).astype(str), rng.randint(1, 6))))
    labels = np.concatenate([labels, ["Zeta", "alpha", "10", "9", "pert_", ""]])   # code-point order, not locale or numeric order
    inp = {"U": labels, "object": labels.astype(object), "series": pd.Series(labels), "series-object": pd.Series(labels.astype(object))}[kind]
    uniq, g = gm.encode_and_count_groups(inp, "non-targeting")
    want_u, want_inv, want_cnt = np.unique(labels, return_inverse=True, return_counts=True)
    assert [str(x) for x in uniq] == [str(x) for x in want_u]
    np.testing.assert_array_equal(g.encoded_groups, want_inv.reshape(-1))
    np.testing.assert_array_equal(g.counts, want_cnt)
    np.testing.assert_array_equal(g.indptr, np.concatenate([[0], np.cumsum(want_cnt)]))
    np.testing.assert_array_equal(g.indices, np.argsort(want_inv.reshape(-1), kind="stable"))
    assert g.encoded_ref_group == int(np.flatnonzero(want_u == "non-targeting")[0])
    for a in (g.encoded_groups, g.counts, g.indices, g.indptr):
        assert a.dtype == np.int64
    assert gm.encode_and_count_groups(inp, None)[1].encoded_ref_group == -1
    with pytest.raises(ValueError, match="is not present"):
        gm.encode_and_count_groups(inp, "no-such-label")
    # integer labels and labels with missing values keep numpy's path (and numpy's behaviour)
    ints = rng.randint(0, 50, size=10000)
    u2, g2 = gm.encode_and_count_groups(ints, 3)
    np.testing.assert_array_equal(u2, np.unique(ints))
    assert g2.encoded_ref_group == 3 and gm._encode_hashed(np.array(["a", None] * 3000, dtype=object), None) is None


def test_product_index_equals_from_product():
    """The (pert, feature) index of the result frame is built with narrow integer codes in one pass; it must be from_product's index
    (asymptotic_wilcoxon.py:252-256): same sorted levels, same codes, same dtypes -- unsorted and duplicated names included."""
    from illico_amd.asymptotic_wilcoxon import _product_index
    rng = np.random.RandomState(2)
    for G, M in [(1, 1), (3, 5), (130, 40), (40, 300)]:
        perts = pd.Series(np.array([f"p{i}" for i in rng.permutation(G)]), name="pert", dtype=str)
        genes = pd.Series(np.array([f"gene_{i}" for i in rng.permutation(M)]), name="feature", dtype=str)
        a = pd.MultiIndex.from_product([perts, genes], names=["pert", "feature"])
        b = _product_index(perts, genes)
        assert a.equals(b) and list(a.names) == list(b.names)
        for x, y in zip(a.levels, b.levels):
            assert x.equals(y) and x.dtype == y.dtype
        for x, y in zip(a.codes, b.codes):
            np.testing.assert_array_equal(x, y)
            assert x.dtype == y.dtype
    dup = pd.Series(["x", "y", "x"], name="feature", dtype=str)
    one = pd.Series(["b", "a"], name="pert", dtype=str)
    assert _product_index(one, dup).equals(pd.MultiIndex.from_product([one, dup], names=["pert", "feature"]))
