"""GPU parity tests (CSC / CSR input) through the C-ABI vs the CPU oracle and the golden vectors."""
import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match, load_golden, make_counts, make_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from illico_amd._lib import get_engine
    return get_engine()


def _run(engine, M, grpc, **kw):
    engine.set_groups(grpc)
    lb, ub = kw.pop("col_lb", 0), kw.pop("col_ub", M.shape[1])
    return engine.run_sparse(M.format, M.data, M.indices, M.indptr, M.shape, lb, ub, **kw)


@pytest.fixture(params=["default", "csr-window", "two-kernel", "two-kernel-sort-only", "csr-regroup", "ovr-lds-sort"])
def route(request, engine):
    """Count-valued CSC genes with groups of at most 255 cells take the LDS-histogram kernel (OVO and OVR); otherwise
    CSC OVO has a single-kernel route (regroup + rank in LDS) with the two-kernel route (regroup into HBM, then
    histogram / sort rank kernels) as fallback.  Count-valued CSR with groups of at most 255 cells takes the group-major single
    pass (k_csr_counts); with that off ("csr-window") or larger groups, dense byte windows + the fused dense kernels; otherwise it is transposed to CSC on the device and takes the CSC routes; the older CSR route
    (regroup by (gene, group) with global atomics) is kept behind an option.  The params force each so that all are
    exercised on the same data.  CSC OVR with values the histogram kernel cannot take (and CSR OVR after the device
    transposition) ranks each gene's stored values inside LDS (k_csc_ovr_gene: value buckets, or a sort of the keys for
    tie-heavy columns); genes larger than its key buffer and the two-kernel params use the general route (regroup in
    HBM, segmented radix sort, sweeps)."""
    opts = {"no_csc_gene_path": 0, "no_dense_window_path": 0, "no_counts_path": 0, "no_csr_transpose_path": 0,
            "no_csr_tile_gather": 0, "no_csc_counts_path": 0, "no_csr_counts_path": 0,
            "no_csc_regroup_lds": 0, "no_csc_ovr_gene_path": 0, "csc_ovr_sorted_form": 0, "no_ovo_ref_buckets": 0}
    if request.param != "default":
        opts.update(no_csr_counts_path=1)
    if request.param.startswith("two-kernel"):
        opts.update(no_csc_gene_path=1, no_dense_window_path=1, no_csc_counts_path=1, no_csc_ovr_gene_path=1)
    if request.param == "ovr-lds-sort":
        # count-valued data too through the single-kernel CSC OVR route, sorted form (sort in LDS + look-ups)
        opts.update(no_dense_window_path=1, no_csc_counts_path=1, csc_ovr_sorted_form=1)
    if request.param == "two-kernel":
        opts.update(no_csr_tile_gather=1)   # CSR -> CSC by the scatter form (what unsorted rows get)
    if request.param.endswith("sort-only"):
        # regroup by k_csc_segment alone, sorted reference column
        opts.update(no_counts_path=1, no_csc_regroup_lds=1, no_ovo_ref_buckets=1)
    if request.param == "csr-regroup":
        opts.update(no_dense_window_path=1, no_csr_transpose_path=1)
    for k, v in opts.items():
        engine.set_option(k, v)
    yield request.param
    for k in opts:
        engine.set_option(k, 0)


@pytest.mark.parametrize("name", ["c1_1k_200_10", "small_ragged", "sparse90", "continuous"])
@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_sparse_matches_reference_goldens(engine, name, fmt, test, route):
    z = load_golden(name)
    X, labels, ref = z["X"], z["labels"], str(z["reference"])
    M = sparse.csc_matrix(X) if fmt == "csc" else sparse.csr_matrix(X)
    from illico_amd.utils.groups import encode_and_count_groups
    keys = [k for k in z.files if k.startswith(f"{fmt}|{test}|") and k.count("|") == 4]
    assert keys
    for key in keys:
        _, _, alt, cc, tc = key.split("|")
        _, g = encode_and_count_groups(labels, ref if test == "ovo" else None)
        got = _run(engine, M, g, use_continuity=bool(int(cc)), tie_correct=bool(int(tc)), alternative=alt)
        gold = z[key]
        assert_planes_match(got, (gold[:, :, 0], gold[:, :, 1], gold[:, :, 2]), ref_row=g.encoded_ref_group,
                            what=f"{name} {key}")


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype,idx", [(np.float32, np.int32), (np.float64, np.int64), (np.int32, np.int32)])
def test_sparse_dtypes_windows_batches(engine, fmt, test, dtype, idx):
    X, rng = make_counts(31, 2500, 150, 0.85)
    labels = make_labels(rng, 2500, 11, n_ref=250)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X.astype(dtype))
    M.indices = M.indices.astype(idx)
    M.indptr = M.indptr.astype(idx)
    want = oracle.run(X.astype(np.float64), g, col_lb=13, col_ub=141)
    engine.set_option("gene_batch", 50)
    try:
        got = _run(engine, M, g, col_lb=13, col_ub=141)
    finally:
        engine.set_option("gene_batch", 0)
    assert_planes_match(got, want, what=f"{fmt} {test} {dtype}")


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_sparse_edge_cases(engine, fmt, test, route):
    """Empty columns, empty rows, an all-dense column, explicit stored zeros (dropped: they are zeros)."""
    rng = np.random.RandomState(7)
    n, m = 1200, 24
    X = rng.poisson(1.0, size=(n, m)).astype(np.float32)
    X[rng.rand(n, m) < 0.7] = 0
    X[:, 0] = 0.0
    X[:, 1] = 1.0 + rng.poisson(3.0, size=n)
    X[5, :] = 0.0
    labels = make_labels(rng, n, 7, n_ref=100)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    # plant explicit zeros in the structure
    M = M.tolil()
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(M)
    M.data[::17] = 0.0
    Xz = M.toarray()
    want = oracle.run(Xz, g)  # dense semantics: a stored zero is a zero
    got = _run(engine, M, g)
    assert_planes_match(got, want, what=f"{fmt} {test} edge")


def test_sparse_device_resident(engine):
    import torch
    X, rng = make_counts(41, 3000, 90, 0.9)
    labels = make_labels(rng, 3000, 10, n_ref=300)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X, g)
    M = sparse.csc_matrix(X)
    engine.set_groups(g)
    d, i, p = (torch.from_numpy(a).cuda() for a in (M.data, M.indices, M.indptr))
    got = engine.run_sparse("csc", d, i, p, M.shape, 0, M.shape[1])
    assert_planes_match(got, want, what="csc device")


def test_csr_sorted_check_and_drop_in(engine):
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    X, rng = make_counts(51, 800, 20, 0.6)
    labels = make_labels(rng, 800, 5)
    M = sparse.csr_matrix(X)
    assert engine.csr_indices_sorted(M.indices, M.indptr, M.shape[0])
    import torch
    assert engine.csr_indices_sorted(torch.from_numpy(M.indices).cuda(), torch.from_numpy(M.indptr).cuda(), M.shape[0])
    bad = M.copy()
    s, e = bad.indptr[3], bad.indptr[4]
    assert e - s >= 2
    bad.indices[s:e] = bad.indices[s:e][::-1].copy()
    assert not engine.csr_indices_sorted(bad.indices, bad.indptr, bad.shape[0])
    assert not engine.csr_indices_sorted(torch.from_numpy(bad.indices).cuda(), torch.from_numpy(bad.indptr).cuda(), bad.shape[0])
    adata = AnnDataLite(bad, obs=pd.DataFrame({"pert": labels}))
    with pytest.raises(ValueError, match="not sorted"):  # reference tests/test_asymptotic_wilcoxon.py:259-273
        asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", reference=labels[0])
    for fmt in ("csc", "csr"):
        A = AnnDataLite((sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X), obs=pd.DataFrame({"pert": labels}))
        df = asymptotic_wilcoxon(A, is_log1p=False, group_keys="pert", reference=labels[0])
        uniq, g = oracle.encode_and_count_groups(labels, labels[0])
        want = oracle.run(X, g)
        got = df.values.reshape(len(uniq), X.shape[1], 3)
        assert_planes_match((got[:, :, 0], got[:, :, 1], got[:, :, 2]), want, ref_row=g.encoded_ref_group, what=fmt)


@pytest.mark.parametrize("fmt", ["csc", "csr"])
def test_sparse_ovo_big_groups_any_values(engine, fmt):
    """Group / reference sizes beyond the in-LDS sort route with non-count values: global radix-sort fallback."""
    rng = np.random.RandomState(17)
    sizes = [30000, 2500, 1200, 300, 5]
    codes = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)])
    rng.shuffle(codes)
    labels = np.array([f"g{c:02d}" for c in codes])
    n = codes.size
    X = (rng.rand(n, 10) * (rng.rand(n, 10) < 0.4)).astype(np.float32)   # continuous positives, 60 % zeros
    X[:, 1] = np.where(rng.rand(n) < 0.5, 0, rng.randint(1, 5, size=n))  # count-valued gene (histogram route)
    X[:, 2] = np.where(rng.rand(n) < 0.7, 0, rng.randn(n))               # negatives stored explicitly
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    for ref in ("g00", "g02"):
        _, g = oracle.encode_and_count_groups(labels, ref)
        want = oracle.run(X, g)     # dense semantics (negatives rank below the zero block)
        got = _run(engine, M, g)
        assert_planes_match(got, want, what=f"{fmt} ref={ref}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_csc_interleaved_dense_and_sparse_genes(engine, test):
    """Every other gene is too dense for the single-kernel CSC route (runs above 128 stored values): the stragglers
    are batched as a column list through the two-kernel route; medium runs (33..128) use the LDS-history form."""
    rng = np.random.RandomState(23)
    sizes = [400, 900, 300, 200, 60, 40]
    codes = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)])
    rng.shuffle(codes)
    labels = np.array([f"g{c:02d}" for c in codes])
    n, m = codes.size, 41
    X = np.zeros((n, m), np.float32)
    for j in range(m):
        dens = (0.9, 0.05, 0.3)[j % 3]          # dense gene, very sparse gene, medium gene
        X[:, j] = rng.poisson(2.0, size=n) * (rng.rand(n) < dens)
        if j % 5 == 0:
            X[:, j] *= rng.rand(n)              # non-count values
    M = sparse.csc_matrix(X)
    _, g = oracle.encode_and_count_groups(labels, "g00" if test == "ovo" else None)
    want = oracle.run(X, g)
    got = _run(engine, M, g)
    assert_planes_match(got, want, what=f"interleaved {test}")
    got = _run(engine, M, g, col_lb=3, col_ub=38)
    want = oracle.run(X, g, col_lb=3, col_ub=38)
    assert_planes_match(got, want, what=f"interleaved window {test}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int64])
@pytest.mark.parametrize("f32_cells", [0, 1])
def test_csr_dense_window_route(engine, test, dtype, f32_cells):
    """CSR through dense windows (byte cells by default, float32 cells behind an option) + the fused kernels: a window wider than one LDS row block (8192 columns),
    windows split by gene_batch, and genes the fused kernels must hand back to the exact sparse route
    (fractional values, a value float32 cannot hold, values beyond the small-integer table, negatives)."""
    rng = np.random.RandomState(101)
    n, m = 260, 8300
    X = (rng.poisson(1.5, size=(n, m)) * (rng.rand(n, m) < 0.12)).astype(np.float64)
    bad_cols = [5, 4100, 8250]
    if dtype != np.int64:
        X[:, 5] = np.where(rng.rand(n) < 0.5, 0.0, rng.rand(n))                 # fractional
        X[3, 4100] = 1.0 + 2.0 ** -30 if dtype == np.float64 else 1.5           # not a float32 integer
    else:
        X[:, 5] = np.where(rng.rand(n) < 0.5, 0, rng.randint(-3, 4, size=n))    # negatives
        X[3, 4100] = 2 ** 40 + 1                                                # beyond float32's integers
    X[7, 8250] = 77                                                             # beyond the table (>= 64)
    labels = make_labels(rng, n, 6, n_ref=40)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    X = X.astype(dtype).astype(np.float64)   # the values the engine sees (float32 rounds the fractional gene)
    M = sparse.csr_matrix(X.astype(dtype))
    want = oracle.run(X, g)
    engine.set_option("dense_window_f32", f32_cells)
    engine.set_option("no_csr_counts_path", 1)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, M, g)
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
        engine.set_option("dense_window_f32", 0)
    assert "k_ovo_fused" in prof or "k_ovr_fused" in prof or "k_group_value_hists" in prof, prof   # the dense-window route ran
    assert_planes_match(got, want, what=f"csr dense window {test} {dtype.__name__}")
    engine.set_option("gene_batch", 3000)
    try:
        got = _run(engine, M, g, col_lb=100, col_ub=8290)
    finally:
        engine.set_option("gene_batch", 0)
        engine.set_option("no_csr_counts_path", 0)
    assert_planes_match(got, oracle.run(X, g, col_lb=100, col_ub=8290), what=f"csr dense window batches {test}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype,idx", [(np.float32, np.int32), (np.float64, np.int64), (np.int64, np.int32), (np.int32, np.int64)])
def test_csr_counts_route(engine, test, dtype, idx):
    """Count-valued CSR through the group-major single pass (k_csr_row_bounds, k_csr_hist, k_csr_tables, k_csr_counts): more genes
    than one window of 2048, groups of 255 cells and of one cell, an empty row, explicit stored zeros, a 4-bit cell that overflows
    (16 cells of one group with the value 9), and genes that must leave the route (a fractional value, a value of 64+, a negative);
    column windows that start and end inside a gene window."""
    rng = np.random.RandomState(503)
    sizes = [255, 255, 120, 61, 33, 16, 2, 1, 700, 256]   # (two groups above 255 cells: chunked histograms + k_csr_big_sweep)
    labels = np.concatenate([[f"s{i:04d}"] * sz for i, sz in enumerate(sizes)])
    rng.shuffle(labels)
    n, m = labels.size, 4300
    X = (rng.poisson(rng.uniform(0.3, 14.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.15)).astype(np.float64)
    X[11, :] = 0                                                                     # an empty row
    X[labels == "s0002", 70] = np.where(rng.rand(120) < 0.5, 9, X[labels == "s0002", 70])   # ~60 cells of one group hold a 9
    X[:, 2050] = np.where(rng.rand(n) < 0.05, rng.randint(64, 300, size=n), X[:, 2050])  # beyond the table
    if dtype in (np.float32, np.float64):
        X[:, 4111] = np.where(rng.rand(n) < 0.1, 0.5, X[:, 4111])                    # fractional
    X[3, 17] = -2.0                                                                  # negative
    X[:, 23] = 0.0                                                                   # an empty gene
    M = sparse.csr_matrix(X.astype(dtype))
    M.indices = M.indices.astype(idx)
    M.indptr = M.indptr.astype(idx)
    M.data[::97] = 0                                                                 # explicit stored zeros ...
    Xd = M.toarray().astype(np.float64)                                              # ... are zeros
    for ref in (["s0000", "s0004", "s0007", "s0008"] if test == "ovo" else [None]):
        _, g = oracle.encode_and_count_groups(labels, ref)
        engine.set_option("profile", 1)
        engine.profile_reset()
        try:
            got = _run(engine, M, g)
            prof = engine.profile_get()
        finally:
            engine.set_option("profile", 0)
        assert "k_csr_counts" in prof, prof
        assert_planes_match(got, oracle.run(Xd, g), what=f"csr counts {test} ref={ref}")
        for lb, ub, kw in [(5, 4290, dict(alternative="less")), (2040, 2060, dict(use_continuity=False)), (100, 2148, dict(tie_correct=False, alternative="greater"))]:
            got = _run(engine, M, g, col_lb=lb, col_ub=ub, **kw)
            assert_planes_match(got, oracle.run(Xd, g, col_lb=lb, col_ub=ub, **kw), what=f"csr counts window {lb}:{ub} {test}")
    # rows out of order: found on the device, every gene goes to the routes that do not need the order
    for r in range(0, n, 7):
        a, b = M.indptr[r], M.indptr[r + 1]
        perm = rng.permutation(b - a)
        M.indices[a:b] = M.indices[a:b][perm]
        M.data[a:b] = M.data[a:b][perm]
    assert_planes_match(_run(engine, M, g), oracle.run(Xd, g), what=f"csr counts {test}, rows out of order")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype,idx", [(np.float32, np.int32), (np.float64, np.int64)])
def test_csr_transposition_route(engine, test, dtype, idx):
    """CSR with continuous values: transposed to CSC on the device (count pass + tile gather), then the CSC routes.
    A window wider than one transposition pass (32768 columns), a column window, rows longer than the register
    window, empty rows, and rows with UNSORTED column indices (those take the scatter form of pass 2)."""
    rng = np.random.RandomState(211)
    n, m = 150, 33100
    X = np.where(rng.rand(n, m) < 0.08, np.log1p(rng.poisson(3.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m))), 0.0)
    X[5, :] = 0.0                                  # an empty row
    X[9, :2000] = rng.rand(2000) + 0.1             # a long dense stretch in one row
    X = X.astype(dtype).astype(np.float64)
    labels = make_labels(rng, n, 5, n_ref=30)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = sparse.csr_matrix(X.astype(dtype))
    M.indices = M.indices.astype(idx)
    M.indptr = M.indptr.astype(idx)
    want = oracle.run(X, g)
    got = _run(engine, M, g)
    assert_planes_match(got, want, what=f"csr transposition {test}")
    got = _run(engine, M, g, col_lb=777, col_ub=32999)
    assert_planes_match(got, oracle.run(X, g, col_lb=777, col_ub=32999), what=f"csr transposition window {test}")
    for r in range(n):                             # shuffle inside every row
        s, e = M.indptr[r], M.indptr[r + 1]
        perm = rng.permutation(e - s)
        M.indices[s:e] = M.indices[s:e][perm]
        M.data[s:e] = M.data[s:e][perm]
    got = _run(engine, M, g)
    assert_planes_match(got, want, what=f"csr transposition, unsorted rows {test}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("many_groups", [False, True])
def test_csc_counts_route(engine, test, many_groups):
    """Count-valued CSC through the LDS-histogram kernel: groups above 255 cells (32-bit rows), a reference above and
    below 255 cells, genes that leave the route (a value of 64+, a fractional value, a negative), explicit stored
    zeros; with 2600 groups the 64-value table does not fit LDS and the 32-value form runs."""
    rng = np.random.RandomState(307)
    if many_groups:
        sizes = [300, 400] + [3] * 2598
    else:
        sizes = [700, 300, 256, 255, 120, 60, 9, 1]
    labels = np.concatenate([[f"s{i:04d}"] * sz for i, sz in enumerate(sizes)])
    rng.shuffle(labels)
    n, m = labels.size, 40
    X = (rng.poisson(rng.uniform(0.3, 12.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.3)).astype(np.float32)
    X[:, 4] = np.where(rng.rand(n) < 0.05, rng.randint(40, 200, size=n), X[:, 4])   # values beyond either table
    X[:, 11] = np.where(rng.rand(n) < 0.1, 0.5, X[:, 11])                            # fractional
    X[3, 17] = -2.0                                                                  # negative
    X[:, 23] = 0.0                                                                   # empty gene
    M = sparse.csc_matrix(X)
    M.data[::97] = 0.0                                                               # explicit stored zeros ...
    Xd = M.toarray()                                                                 # ... are zeros
    for ref in (["s0000", "s0004"] if test == "ovo" else [None]):
        _, g = oracle.encode_and_count_groups(labels, ref)
        want = oracle.run(Xd, g)
        engine.set_option("profile", 1)
        engine.profile_reset()
        try:
            got = _run(engine, M, g)
            prof = engine.profile_get()
        finally:
            engine.set_option("profile", 0)
        assert "k_csc_counts" in prof, prof
        assert_planes_match(got, want, what=f"csc counts {test} ref={ref} many_groups={many_groups}")
        got = _run(engine, M, g, col_lb=2, col_ub=31)
        assert_planes_match(got, oracle.run(Xd, g, col_lb=2, col_ub=31), what=f"csc counts window {test}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_csc_counts_mixed_cells_overflow_is_redone_with_8bit_cells(engine, test):
    """k_csc_counts first runs with 8-bit cells for the values 1..7 and 4-bit cells for 8..63 (two workgroups per CU).  Here
    most groups hold 16 or more cells with the same value >= 8 in several genes: those 4-bit cells overflow, the gene is
    flagged (the cells no longer add up to the entries counted) and redone with 8-bit cells.  Genes without an overflow
    stay on the mixed form; the planes equal the 8-bit-only run byte for byte and match the oracle."""
    rng = np.random.RandomState(311)
    sizes = [250] * 40 + [254, 255, 17, 16, 15, 1]
    labels = np.concatenate([[f"s{i:03d}"] * sz for i, sz in enumerate(sizes)])
    rng.shuffle(labels)
    n, m = labels.size, 24
    X = np.zeros((n, m), np.float32)
    for j in range(m):
        if j % 3 == 0:     # narrow distribution around 9..11, dense: 4-bit cells overflow in most groups
            X[:, j] = rng.randint(9, 12, size=n) * (rng.rand(n) < 0.7)
        elif j % 3 == 1:   # small values only: the 8-bit cells of the mixed layout take them (up to 255 per cell)
            X[:, j] = rng.randint(1, 4, size=n) * (rng.rand(n) < 0.9)
        else:              # spread-out values, sparse: no cell reaches 16
            X[:, j] = rng.randint(1, 64, size=n) * (rng.rand(n) < 0.1)
    X[:, 5] = 15.0          # exactly 15 / 16 / 17 copies in the groups of 15 / 16 / 17 cells: the overflow boundary
    X[:, 8] = 63.0          # the last 4-bit cell of the last word: its carry leaves the word
    M = sparse.csc_matrix(X)
    _, g = oracle.encode_and_count_groups(labels, "s000" if test == "ovo" else None)
    want = oracle.run(X, g)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, M, g)
        prof = engine.profile_get()
        engine.set_option("no_csc_counts_mixed", 1)
        only8 = _run(engine, M, g)
    finally:
        engine.set_option("no_csc_counts_mixed", 0)
        engine.set_option("profile", 0)
    assert prof["k_csc_counts"]["launches"] == 2, prof   # the mixed pass, then the 8-bit pass over the flagged genes
    assert set(prof) <= {"k_csc_counts", "k_finalize", "k_gene_totals"}, prof   # no gene left the histogram route
    for a, b in zip(got, only8):
        assert a.tobytes() == b.tobytes()
    assert_planes_match(got, want, ref_row=g.encoded_ref_group if test == "ovo" else None, what=f"mixed cells {test}")


def test_drop_in_csr_narrow_dtypes(engine):
    """In-RAM CSR goes to the device inside the drop-in call: value / index dtypes the kernels do not take natively
    (uint16 counts, int64 indptr with int32 indices) are widened on the way like on the host path."""
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    X, rng = make_counts(77, 900, 33, 0.7)
    labels = make_labels(rng, 900, 6)
    M = sparse.csr_matrix(X.astype(np.uint16))
    M.indptr = M.indptr.astype(np.int64)
    df = asymptotic_wilcoxon(AnnDataLite(M, obs=pd.DataFrame({"pert": labels})), is_log1p=False, group_keys="pert", reference=labels[0])
    uniq, g = oracle.encode_and_count_groups(labels, labels[0])
    want = oracle.run(X, g)
    got = df.values.reshape(len(uniq), X.shape[1], 3)
    assert_planes_match((got[:, :, 0], got[:, :, 1], got[:, :, 2]), want, ref_row=g.encoded_ref_group, what="csr uint16")


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("dtype,idx", [(np.float32, np.int32), (np.float64, np.int64), (np.int32, np.int32)])
@pytest.mark.parametrize("sorted_form", [0, 1])
def test_csc_ovr_single_kernel_route(engine, fmt, dtype, idx, sorted_form):
    """OVR, any values, single kernel per gene (k_csc_ovr_gene).  Spread-out values are ranked inside value buckets,
    tie-heavy columns (and everything with sorted_form = 1) by sorting the keys in LDS: columns whose stored entries fill
    1, 3, 5 and 30+ sort chunks of 1024 keys (LDS merge levels over a chunk count that is not a power of two), negatives
    stored explicitly, tie-heavy and all-distinct columns, explicit stored zeros, an empty column, a constant column,
    and columns with more stored entries than the LDS key buffer holds (those fall back to the general route as a
    column list)."""
    rng = np.random.RandomState(401)
    n, m = 52000, 14
    sizes = [20000, 9000, 700, 300, 255, 40, 3, 1]
    sizes.append(n - sum(sizes))
    labels = np.concatenate([[f"s{i:02d}"] * sz for i, sz in enumerate(sizes)])
    rng.shuffle(labels)
    dens = [0.01, 0.05, 0.09, 0.58, 0.97, 0.0, 0.3, 0.02, 0.2, 0.66, 0.5, 0.0007, 0.058, 1.0]
    X = np.zeros((n, m))
    for j in range(m):
        keep = rng.rand(n) < dens[j]
        if dtype == np.int32:
            vals = rng.randint(-3, 90, size=n) if j % 2 else rng.randint(1, 4, size=n)
        elif j % 3 == 0:
            vals = np.log1p(rng.poisson(3.0, size=n) * rng.uniform(0.5, 1.5, size=n))   # continuous, some exact repeats (0)
        elif j % 3 == 1:
            vals = rng.randn(n)                                                         # negatives
        else:
            vals = rng.poisson(2.0, size=n).astype(np.float64) + 0.5 * (rng.rand(n) < 0.3)  # tie-heavy
        X[:, j] = np.where(keep, vals, 0.0)
    X[:, 6] = np.where(X[:, 6] != 0, 2.5 if dtype != np.int32 else 2, 0)                 # one value only
    X = X.astype(dtype).astype(np.float64)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X.astype(dtype))
    M.indices = M.indices.astype(idx)
    M.indptr = M.indptr.astype(idx)
    M.data[::53] = 0                                                                     # explicit stored zeros
    Xd = M.toarray().astype(np.float64)
    _, g = oracle.encode_and_count_groups(labels, None)
    want = oracle.run(Xd, g)
    engine.set_option("no_dense_window_path", 1)
    engine.set_option("no_csc_counts_path", 1)
    engine.set_option("csc_ovr_sorted_form", sorted_form)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, M, g)
        prof = engine.profile_get()
        got_w = _run(engine, M, g, col_lb=2, col_ub=13, alternative="greater", use_continuity=False)
    finally:
        engine.set_option("profile", 0)
        engine.set_option("csc_ovr_sorted_form", 0)
        engine.set_option("no_dense_window_path", 0)
        engine.set_option("no_csc_counts_path", 0)
    assert "k_csc_ovr_gene" in prof and "k_ovr_gene" in prof, prof   # the new route ran, and the oversized columns fell back
    assert_planes_match(got, want, what=f"csc ovr single kernel {fmt} {dtype.__name__} sorted_form={sorted_form}")
    want_w = oracle.run(Xd, g, col_lb=2, col_ub=13, alternative="greater", use_continuity=False)
    assert_planes_match(got_w, want_w, what=f"csc ovr single kernel window {fmt} {dtype.__name__}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("fmt", ["csc", "csr"])
def test_explicitly_stored_zeros_are_ranked_as_zeros(engine, test, fmt):
    """A deliberate difference from the reference, pinned on its own outputs (tests/golden/stored_zeros.npz): for sparse input
    with explicitly stored zeros the engine returns what the reference's DENSE kernels return for the same numbers; the
    reference's sparse kernels rank a stored zero above the implicit zeros and give different statistics throughout."""
    z = load_golden("stored_zeros")
    labels, ref = z["labels"], str(z["reference"])
    from illico_amd.utils.groups import encode_and_count_groups
    _, g = encode_and_count_groups(labels, ref if test == "ovo" else None)
    M = sparse.csc_matrix((z["csc_data"], z["csc_indices"], z["csc_indptr"]), shape=z["X"].shape)
    if fmt == "csr":
        Mr = sparse.csr_matrix(z["Xz"])          # CSR with the same explicit zeros: rebuild from the CSC entries
        coo = M.tocoo(copy=True)
        Mr = sparse.csr_matrix((coo.data, (coo.row, coo.col)), shape=M.shape)
        M = Mr
    assert (M.data == 0).sum() > 50
    got = _run(engine, M, g)
    dense, sparse_ref = z[f"dense|{test}"], z[f"csc|{test}"]
    assert_planes_match(got, (dense[:, :, 0], dense[:, :, 1], dense[:, :, 2]), ref_row=g.encoded_ref_group if test == "ovo" else None,
                        what=f"stored zeros {fmt} {test}")
    rows = np.arange(dense.shape[0]) != (g.encoded_ref_group if test == "ovo" else -1)
    assert (got[1][rows] != sparse_ref[rows][:, :, 1]).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_sparse_type_extremes(engine, fmt, test, dtype, route):
    """The largest / smallest value of the type (the largest integer's sortable key is the all-ones pattern the kernels use for
    empty slots and padding), +-infinity: a few cells of a column, a third of a column.  Negative stored values are ranked with
    dense semantics, so the dense oracle is the judge."""
    rng = np.random.RandomState(3)
    n, m = 3000, 8
    labels = make_labels(rng, n, 25, n_ref=600)
    integer = np.issubdtype(dtype, np.integer)
    big = np.iinfo(dtype).max if integer else np.inf
    small = np.iinfo(dtype).min if integer else -np.inf
    X = (rng.poisson(3.0, size=(n, m)) * rng.randint(1, 1000, size=(n, m))).astype(dtype)
    X[rng.rand(n, m) < 0.7] = 0
    X[rng.rand(n) < 0.01, 0] = big
    X[rng.rand(n) < 0.3, 1] = big
    X[rng.rand(n) < 0.05, 2] = small
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(np.ascontiguousarray(X, dtype=np.float64), g)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    got = _run(engine, M, g)
    assert_planes_match(got, want, ref_row=g.encoded_ref_group if test == "ovo" else None, what=f"{fmt} {test} {np.dtype(dtype).name} {route}")


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_deferred_csc_calls_complete_their_leftover_genes(engine, test, fmt):
    """ILLICO_FLAG_DEFER on device-resident CSC arrays (and CSR arrays: the group-major pass, groups above 255 cells included): the count-valued pass is enqueued without a host wait; the genes it
    cannot take (values beyond the table, fractional values, 4-bit cells that overflow) are recomputed when the next call or
    synchronize() looks at their flags -- and a matrix that is no count matrix at all is sent on from the device-side sample."""
    import torch
    X, rng = make_counts(31, 6000, 96, 0.85)
    X[:, 7] *= 9.0                          # values beyond 63 in one gene
    X[::3, 40] += 0.5                       # fractional values in another
    X[:, 41] = np.where(rng.rand(6000) < 0.9, 9.0, 0.0)  # sixteen and more equal values of 9 per group: the 4-bit cells overflow
    labels = make_labels(rng, 6000, 9, n_ref=400)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(X, g)
    Xc, rng2 = make_counts(32, 6000, 96, 0.85)
    Xc = np.log1p(Xc * rng2.uniform(0.5, 1.5, size=Xc.shape)).astype(np.float32)  # continuous: the sample sends every gene on
    want_c = oracle.run(Xc, g)
    engine.set_groups(g)
    dev = torch.device("cuda", engine.device)

    def up(M):
        M = sparse.csc_matrix(M) if fmt == "csc" else sparse.csr_matrix(M)
        return tuple(torch.from_numpy(a).to(dev) for a in (M.data, M.indices, M.indptr)), M.shape

    (d, i, p), shape = up(X)
    (dc, ic, pc), _ = up(Xc)
    G = g.counts.size
    A = tuple(torch.full((G, 96), -7.0, dtype=torch.float64, device=dev) for _ in range(3))
    B = tuple(torch.full((G, 96), -7.0, dtype=torch.float64, device=dev) for _ in range(3))
    engine.run_sparse(fmt, d, i, p, shape, 0, 96, out=A, defer=True)
    engine.run_sparse(fmt, dc, ic, pc, shape, 0, 96, out=B, defer=True)   # other planes: enqueued before A is completed
    engine.run_sparse(fmt, d, i, p, shape, 0, 96, out=A, defer=True)      # the same planes again
    engine.synchronize()
    ref_row = g.encoded_ref_group
    assert_planes_match(tuple(t.cpu().numpy() for t in A), want, ref_row=ref_row, what=f"deferred {fmt} {test}")
    assert_planes_match(tuple(t.cpu().numpy() for t in B), want_c, ref_row=ref_row, what=f"deferred {fmt} {test} continuous")
    # a window, then a non-deferred call while one is pending
    C = tuple(torch.full((G, 96), -7.0, dtype=torch.float64, device=dev) for _ in range(3))
    engine.run_sparse(fmt, d, i, p, shape, 32, 80, out=tuple(t[:, 32:80] for t in C), defer=True)
    got = engine.run_sparse(fmt, d, i, p, shape, 0, 96)
    assert_planes_match(got, want, ref_row=ref_row, what=f"{fmt} {test} after a deferred call")
    engine.synchronize()
    assert_planes_match(tuple(t[:, 32:80].cpu().numpy() for t in C), tuple(w[:, 32:80] for w in want), ref_row=ref_row,
                        what=f"deferred {fmt} window {test}")
    assert all(float(t[:, :32].min()) == -7.0 and float(t[:, 80:].max()) == -7.0 for t in C)


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_csc_counts_large_groups_with_one_dominant_value(engine, test):
    """Groups of tens of thousands of cells whose stored values are nearly all the same count: the per-value tie terms
    3 tS (tS + tB) + tB^2 of the 32-bit-cell rows leave 32 bits (25 000 reference cells and 30 000 group cells with the value 1:
    5.0e9) -- they are formed in 64 bits."""
    rng = np.random.RandomState(5)
    sizes = [25000, 30000, 4000, 700, 200, 90]
    codes = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(codes)
    n = codes.size
    labels = np.array(["non-targeting" if c == 0 else f"pert_{c:05d}" for c in codes])
    X = np.zeros((n, 6), dtype=np.float32)
    X[:, 0] = 1.0                                   # every cell the same count
    X[:, 1] = (rng.rand(n) < 0.97).astype(np.float32)   # nearly every cell
    X[:, 2] = rng.poisson(0.3, size=n)
    X[:, 3] = np.where(rng.rand(n) < 0.9, 7.0, 0.0)
    X[:, 4] = rng.poisson(6.0, size=n)
    X[:, 5] = np.where(codes == 1, 3.0, 1.0)        # one value per group
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = sparse.csc_matrix(X)
    got = _run(engine, M, g)
    assert_planes_match(got, oracle.run(X, g), ref_row=g.encoded_ref_group, what=f"csc large groups {test}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("n_groups", [14, 1300])
def test_csc_counts_many_groups_above_255_cells_take_16_bit_cells(engine, test, n_groups):
    """More groups above 255 cells than the kernel's side table of 32-bit rows holds (8): the LDS-histogram kernel keeps 16-bit cells
    for EVERY group (128 bytes per group: in one launch up to ~1180 groups, in windows of groups beyond: 1300 groups) instead of leaving the window to the
    per-gene LDS sort (19 ms against 0.6 ms at 300k x 8k x 300).  Host CSC, and device arrays deferred; genes with values beyond the
    table, fractional values and explicit zeros among them; identical to the other route (`no_csc_counts_wide`)."""
    import torch
    rng = np.random.RandomState(41)
    sizes = [900] + [int(s) for s in rng.randint(260, 700 if n_groups < 100 else 290, size=n_groups - 3)] + [40, 3]
    codes = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(codes)
    n, m = codes.size, 48
    labels = np.array(["non-targeting" if c == 0 else f"pert_{c:05d}" for c in codes])
    X = (rng.poisson(rng.uniform(0.3, 9.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.25)).astype(np.float32)
    X[:, 5] = rng.poisson(25.0, size=n) * (rng.rand(n) < 0.3)          # values in [32, 64)
    X[:, 9] = rng.poisson(80.0, size=n) * (rng.rand(n) < 0.3)          # beyond every table: another route
    X[:, 11] = X[:, 11] * 0.5                                          # fractional
    X[:, 13] = 1.0                                                     # every cell the same count
    X[:, 17] = 0.0
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(X, g)
    M = sparse.csc_matrix(X)
    M.data[::97] = 0.0                                                 # explicit zeros
    Xz = M.toarray()
    want = oracle.run(Xz, g)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_sparse("csc", M.data, M.indices, M.indptr, M.shape, 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_csc_counts" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"csc 16-bit cells {test} G={n_groups}")
    dev = torch.device("cuda", engine.device)
    d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (M.data, M.indices, M.indptr))
    out = tuple(torch.full((g.counts.size, m), -7.0, dtype=torch.float64, device=dev) for _ in range(3))
    engine.run_sparse("csc", d, i, p, M.shape, 0, m, out=out, defer=True)
    engine.synchronize()
    assert_planes_match(tuple(t.cpu().numpy() for t in out), want, ref_row=g.encoded_ref_group, what=f"csc 16-bit cells deferred {test} G={n_groups}")
    engine.set_option("no_csc_counts_wide", 1)
    try:
        engine.profile(True)
        engine.profile_reset()
        again = engine.run_sparse("csc", M.data, M.indices, M.indptr, M.shape, 0, m)
        prof2 = engine.profile_get()
    finally:
        engine.profile(False)
        engine.set_option("no_csc_counts_wide", 0)
    assert "k_csc_counts" not in prof2, prof2
    # (columns of 90 000 stored entries: without the histogram kernel the window is written out dense and takes the dense routes, whose OVR
    #  tie sum is the exact integer where the sparse routes follow the reference's float64 accumulation -- kernels_finalize.h: tie_f64_sparse;
    #  p then agrees to the last few bits, the statistic and the fold change exactly)
    for k, (a, b) in enumerate(zip(got, again)):
        if k == 0 and test == "ovr":
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=0.0)
        else:
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_csc_counts_more_groups_than_lds_tables_are_taken_in_windows(engine, test):
    """6000 groups of a few cells: their histogram tables do not fit LDS at once (36 bytes per group); the LDS-histogram kernel takes
    them in windows of ~2000 groups, one launch each over the same entries (a genome-wide screen's shape: the route used to fall back
    to the per-gene sort, 44 ms against 3.8 ms at 300k x 8k x 10000).  Host CSC and device arrays deferred; identical to the other
    route (`no_csc_counts_windows`)."""
    import torch
    rng = np.random.RandomState(43)
    G, m = 6000, 24
    sizes = np.concatenate([[700], rng.randint(1, 9, size=G - 2), [300]])   # the reference, small groups, one group above 255 cells
    codes = np.repeat(np.arange(G), sizes)
    rng.shuffle(codes)
    n = codes.size
    labels = np.array(["non-targeting" if c == 0 else f"pert_{c:05d}" for c in codes])
    X = (rng.poisson(rng.uniform(0.3, 7.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.3)).astype(np.float32)
    X[:, 3] = rng.poisson(40.0, size=n) * (rng.rand(n) < 0.3)    # up to ~63: 4-bit cells overflow -> the 8-bit form, in windows too
    X[:, 7] = rng.poisson(90.0, size=n) * (rng.rand(n) < 0.2)    # beyond the table: another route
    X[:, 9] = 2.0
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(X, g)
    M = sparse.csc_matrix(X)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_sparse("csc", M.data, M.indices, M.indptr, M.shape, 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert prof.get("k_csc_counts", {}).get("launches", 0) >= 3, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"csc group windows {test}")
    dev = torch.device("cuda", engine.device)
    d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (M.data, M.indices, M.indptr))
    out = tuple(torch.full((g.counts.size, m), -7.0, dtype=torch.float64, device=dev) for _ in range(3))
    engine.run_sparse("csc", d, i, p, M.shape, 0, m, out=out, defer=True)
    engine.synchronize()
    assert_planes_match(tuple(t.cpu().numpy() for t in out), want, ref_row=g.encoded_ref_group, what=f"csc group windows deferred {test}")
    engine.set_option("no_csc_counts_windows", 1)
    try:
        again = engine.run_sparse("csc", M.data, M.indices, M.indptr, M.shape, 0, m)
    finally:
        engine.set_option("no_csc_counts_windows", 0)
    for a, b in zip(got, again):
        np.testing.assert_array_equal(a, b)


def test_csc_counts_with_a_reference_of_tens_of_thousands_of_cells(engine):
    """An OVO reference of 30 000 cells or more (a tenth of a large screen's cells are controls): the sweep's 32-bit terms (3 tS^2)
    would overflow, so the LDS-histogram kernel takes its 16-bit-cell form, whose terms are 64-bit, instead of leaving the window to
    the per-gene sort.  Most reference cells share one value in some genes (tS ~ 40 000)."""
    rng = np.random.RandomState(47)
    sizes = [45000] + [int(s) for s in rng.randint(40, 400, size=60)]
    codes = np.repeat(np.arange(len(sizes)), sizes)
    rng.shuffle(codes)
    n, m = codes.size, 16
    labels = np.array(["non-targeting" if c == 0 else f"pert_{c:05d}" for c in codes])
    X = (rng.poisson(rng.uniform(0.3, 6.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.3)).astype(np.float32)
    X[:, 2] = 1.0                                        # tS = 45 000
    X[:, 5] = (rng.rand(n) < 0.95) * 3.0
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X, g)
    M = sparse.csc_matrix(X)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_sparse("csc", M.data, M.indices, M.indptr, M.shape, 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_csc_counts" in prof and "k_csc_gene" not in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what="csc counts, large reference")


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sparse_ovo_with_groups_of_thousands_of_cells_takes_the_packed_rank_kernel(engine, fmt, dtype):
    """Continuous sparse OVO with cluster-sized groups: the regrouped (gene, group) runs go to the dense route's packed rank kernel
    (runs above 256 keys dealt into value buckets and walked in pieces); a count-valued gene takes the histogram kernel, a gene whose
    reference run is tie-heavy and one with a value repeated 500 times in one group the general routes.  Against the oracle."""
    rng = np.random.RandomState(4242)
    sizes = [2500, 4000, 1800, 1100, 600, 300, 257, 33, 1]   # group 0: the reference
    labels = np.concatenate([["non-targeting"] * sizes[0]] + [[f"c{i:02d}"] * sz for i, sz in enumerate(sizes[1:])])
    rng.shuffle(labels)
    n, m = labels.size, 40
    X = np.where(rng.rand(n, m) < 0.3, np.round(np.log1p(rng.poisson(4.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m))), 3), 0.0)
    X[:, 3] = np.where(rng.rand(n) < 0.4, np.round(rng.rand(n) * 3, 1) + 0.1, 0.0)                 # ~30 distinct values
    X[labels == "c00", 7] = np.where(rng.rand(4000) < 0.15, 1.234, X[labels == "c00", 7])          # one value ~600 times in a big group
    X[:, 11] = rng.poisson(2.0, size=n) * (rng.rand(n) < 0.3)                                      # counts
    X[:, 13] = 0.0                                                                                 # an empty gene
    X = X.astype(dtype)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X.astype(np.float64) if dtype == np.float64 else X, g)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, M, g)
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
    assert "k_ovo_rank_compact" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"sparse OVO, big groups {fmt} {np.dtype(dtype).name}")
    got = _run(engine, M, g, col_lb=2, col_ub=30, alternative="less")
    assert_planes_match(got, oracle.run(X, g, col_lb=2, col_ub=30, alternative="less"), ref_row=g.encoded_ref_group, what="sparse OVO, big groups, window")
    engine.set_option("no_packed_dense", 1)   # ... and the routes it replaces give the same planes
    try:
        old = _run(engine, M, g)
    finally:
        engine.set_option("no_packed_dense", 0)
    for a, b in zip(_run(engine, M, g), old):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("where", ["host", "device"])
def test_bound_csr_chunk_calls_are_served_from_windows_computed_ahead(engine, test, where):
    """`bound_ahead_genes`: the reference's driver asks a dispatcher for ~256 genes at a time (asymptotic_wilcoxon.py:213-241), and a CSR
    call walks every row whatever its width.  With the option set, a narrow call on a bound CSR matrix computes the aligned window
    around it once and later calls are slices of it: chunks in a scrambled order, a chunk that straddles two windows, the last
    (short) window, other flags (their own windows), new groups (the windows are dropped), against the oracle every time."""
    import torch
    rng = np.random.RandomState(811)
    n, m = 3000, 1500
    X = (rng.poisson(rng.uniform(0.3, 9.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.2)).astype(np.float32)
    labels = make_labels(rng, n, 12, n_ref=200)
    M = sparse.csr_matrix(X)
    dev = torch.device("cuda", engine.device)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    engine.set_groups(g)
    bm = engine.bind_sparse("csr", M.data, M.indices, M.indptr, M.shape)
    G = g.counts.size

    def chunked(bounds, **kw):
        if where == "host":
            out = tuple(np.full((G, m), -7.0) for _ in range(3))
            for lb, ub in bounds:
                got = bm.run(lb, ub, **kw)
                for k in range(3):
                    out[k][:, lb:ub] = got[k]
            return out
        out = tuple(torch.full((G, m), -7.0, dtype=torch.float64, device=dev) for _ in range(3))
        for lb, ub in bounds:
            bm.run(lb, ub, out=tuple(t[:, lb:ub] for t in out), **kw)
        engine.synchronize()
        return tuple(t.cpu().numpy() for t in out)

    engine.set_option("bound_ahead_genes", 512)
    try:
        bounds = [(lb, min(lb + 100, m)) for lb in range(0, m, 100)]     # 100-gene chunks: some straddle a 512 boundary
        engine.set_option("profile", 1)
        engine.profile_reset()
        got = chunked(bounds)
        prof = engine.profile_get()
        engine.set_option("profile", 0)
        assert_planes_match(got, oracle.run(X, g), ref_row=g.encoded_ref_group, what=f"ahead {test} {where}")
        assert prof["k_csr_counts"]["launches"] <= 2 * 3, prof["k_csr_counts"]   # three windows were computed ([0, 512), [500, 1012), [1000, 1500)), not 15 chunks
        got = chunked([bounds[i] for i in rng.permutation(len(bounds))])
        assert_planes_match(got, oracle.run(X, g), ref_row=g.encoded_ref_group, what=f"ahead {test} {where}, scrambled order")
        kw = dict(alternative="less", use_continuity=False)
        got = chunked(bounds, **kw)
        assert_planes_match(got, oracle.run(X, g, **kw), ref_row=g.encoded_ref_group, what=f"ahead {test} {where} other flags")
        # new groups: a window computed for the old ones must not be handed out
        labels2 = make_labels(rng, n, 7, n_ref=300)
        _, g2 = oracle.encode_and_count_groups(labels2, "non-targeting" if test == "ovo" else None)
        engine.set_groups(g2)
        G = g2.counts.size
        got = chunked(bounds)
        assert_planes_match(got, oracle.run(X, g2), ref_row=g2.encoded_ref_group, what=f"ahead {test} {where} new groups")
        # a call at least as wide as the window goes straight through
        got = bm.run(0, m)
        assert_planes_match(got, oracle.run(X, g2), ref_row=g2.encoded_ref_group, what=f"ahead {test} {where} whole")
    finally:
        engine.set_option("profile", 0)
        engine.set_option("bound_ahead_genes", 0)
        bm.release()


def test_adopted_arrays_rewritten_in_place_are_looked_at_again_after_touch(engine):
    """`illico_matrix_touch` (include/illico_hip.h): a bound matrix that ADOPTED device arrays remembers windows computed ahead and
    the rows' order; the caller rewrites the values in place (a normalisation) and reverses every row's entries (rows no longer in
    order), touches the handle, and gets the new matrix's results -- not slices of the old windows, not a pass that trusts the old order."""
    import torch
    rng = np.random.RandomState(4)
    n, m = 2500, 700
    X = (rng.poisson(rng.uniform(0.3, 9.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.2)).astype(np.float32)
    labels = make_labels(rng, n, 9, n_ref=200)
    M = sparse.csr_matrix(X)
    dev = torch.device("cuda", engine.device)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    engine.set_groups(g)
    d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (M.data, M.indices, M.indptr))
    bm = engine.bind_sparse("csr", d, i, p, M.shape)
    engine.set_option("bound_ahead_genes", 512)
    try:
        got = bm.run(100, 200)
        assert_planes_match(got, tuple(a[:, 100:200] for a in oracle.run(X, g)), ref_row=g.encoded_ref_group, what="adopted, first")
        # in place: other values, and every row's entries in descending column order
        X2 = X.copy()
        X2[X2 > 0] += 3.0
        M2 = sparse.csr_matrix(X2)
        data2, idx2 = M2.data.copy(), M2.indices.copy()
        for r in range(n):
            a, b = M2.indptr[r], M2.indptr[r + 1]
            data2[a:b] = data2[a:b][::-1]
            idx2[a:b] = idx2[a:b][::-1]
        d.copy_(torch.from_numpy(data2).to(dev))
        i.copy_(torch.from_numpy(idx2).to(dev))
        bm.touch()
        want = oracle.run(X2, g)
        got = bm.run(100, 200)
        assert_planes_match(got, tuple(a[:, 100:200] for a in want), ref_row=g.encoded_ref_group, what="adopted, rewritten, chunk")
        got = bm.run(0, m)
        assert_planes_match(got, want, ref_row=g.encoded_ref_group, what="adopted, rewritten, whole")
    finally:
        engine.set_option("bound_ahead_genes", 0)
        bm.release()


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype,idx", [(np.float32, np.int32), (np.float64, np.int64)])
def test_csr_with_long_columns_of_any_values_takes_the_dense_routes(engine, test, dtype, idx):
    """A CSR matrix a third of whose cells are stored, continuous values: a column holds more keys (36 000) than the per-gene LDS
    kernels behind the transposition take, and every gene would fall to the general sort routes (76 ms at C3 shape).  The window is
    written out dense in the matrix's own type (k_csr_densify) and takes the dense routes; same results as the sparse routes
    (`no_csr_densify_any`) within the reference's tolerance, against the oracle."""
    rng = np.random.RandomState(911)
    n, m = 120_000, 70
    X = (np.exp(rng.normal(0.0, 1.0, size=(n, m))) * (rng.rand(n, m) < 0.33)).astype(dtype)
    X[:, 5] = np.round(X[:, 5])                      # a count-like gene among them
    X[:, 9] = 0                                      # an empty gene
    labels = make_labels(rng, n, 25, n_ref=3000)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = sparse.csr_matrix(X)
    M.indices = M.indices.astype(idx)
    M.indptr = M.indptr.astype(idx)
    want = oracle.run(X.astype(np.float64), g)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, M, g)
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
    assert "k_densify" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"dense-ish csr {test}")
    got = _run(engine, M, g, col_lb=3, col_ub=67, alternative="greater")
    assert_planes_match(got, oracle.run(X.astype(np.float64), g, col_lb=3, col_ub=67, alternative="greater"), ref_row=g.encoded_ref_group, what=f"dense-ish csr {test} window")
    engine.set_option("no_csr_densify_any", 1)
    try:
        again = _run(engine, M, g)
    finally:
        engine.set_option("no_csr_densify_any", 0)
    assert_planes_match(again, want, ref_row=g.encoded_ref_group, what=f"dense-ish csr {test}, sparse routes")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype,idx", [(np.float32, np.int32), (np.float64, np.int64)])
def test_csc_with_long_columns_of_any_values_takes_the_dense_routes(engine, test, dtype, idx):
    """The CSC form of the test above (k_csc_densify: the columns' row indices must ascend -- with one column's rows out of order the
    window stays with the sparse routes)."""
    rng = np.random.RandomState(913)
    n, m = 120_000, 70
    X = (np.exp(rng.normal(0.0, 1.0, size=(n, m))) * (rng.rand(n, m) < 0.33)).astype(dtype)
    X[:, 7] = 0
    X[-1, :] = 1.5                                   # the last row is stored in every column (a chunk boundary at the matrix's end)
    labels = make_labels(rng, n, 25, n_ref=3000)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = sparse.csc_matrix(X)
    M.indices = M.indices.astype(idx)
    M.indptr = M.indptr.astype(idx)
    want = oracle.run(X.astype(np.float64), g)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, M, g)
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
    assert "k_densify" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"dense-ish csc {test}")
    got = _run(engine, M, g, col_lb=3, col_ub=67, alternative="less")
    assert_planes_match(got, oracle.run(X.astype(np.float64), g, col_lb=3, col_ub=67, alternative="less"), ref_row=g.encoded_ref_group, what=f"dense-ish csc {test} window")
    a, b = M.indptr[11], M.indptr[12]               # one column's rows out of order
    perm = rng.permutation(b - a)
    M.indices[a:b] = M.indices[a:b][perm]
    M.data[a:b] = M.data[a:b][perm]
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, M, g)
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
    assert "k_densify" not in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"dense-ish csc {test}, a column out of order")


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_float64_sparse_values_that_are_float32_values_take_the_float32_kernels(engine, fmt, test):
    """A float32 matrix widened to float64 on its way (device-resident arrays): every stored value is a float32 value, so the float32
    kernels -- whose per-gene LDS buffers hold twice the keys -- give the same bits: same planes as with `no_f64_narrowing`, and the
    oracle's.  A matrix with ONE value that float32 cannot hold stays with the float64 kernels (same planes again)."""
    import torch
    rng = np.random.RandomState(2024)
    n, m = 30_000, 150
    X = (np.exp(rng.normal(0.0, 1.0, size=(n, m))).astype(np.float32) * (rng.rand(n, m) < 0.1)).astype(np.float64)
    labels = make_labels(rng, n, 40, n_ref=1500)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    dev = torch.device("cuda", engine.device)
    engine.set_groups(g)

    def run(Xh):
        M = (sparse.csr_matrix if fmt == "csr" else sparse.csc_matrix)(Xh)
        d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (M.data, M.indices, M.indptr))
        return engine.run_sparse(fmt, d, i, p, M.shape, 0, m)

    want = oracle.run(X, g)
    got = run(X)
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"f64 holding f32 values {fmt} {test}")
    engine.set_option("no_f64_narrowing", 1)
    try:
        wide = run(X)
    finally:
        engine.set_option("no_f64_narrowing", 0)
    for a, b in zip(got, wide):
        np.testing.assert_array_equal(a, b)
    X2 = X.copy()
    r, c = np.argwhere(X2 != 0)[123]
    X2[r, c] = 1.0 + 2.0 ** -40                     # not a float32 value
    assert_planes_match(run(X2), oracle.run(X2, g), ref_row=g.encoded_ref_group, what=f"f64 with one wide value {fmt} {test}")


@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_float64_sparse_values_with_is_log1p_keep_the_float64_kernels(engine, fmt):
    """is_log1p: the reference forms expm1 of the float64 data in float64 (utils/sparse/csr.py:282, csc.py:207), the float32 kernels
    in float32 -- a fold change 1e-7 apart.  So a float64 matrix that holds float32 values is NOT narrowed when is_log1p is set:
    the planes are the oracle's at rtol 1e-12 and bit for bit those of `no_f64_narrowing`."""
    import torch
    rng = np.random.RandomState(7)
    n, m = 20_000, 96
    X = (np.log1p(np.exp(rng.normal(0.0, 1.0, size=(n, m)))).astype(np.float32) * (rng.rand(n, m) < 0.1)).astype(np.float64)
    labels = make_labels(rng, n, 30, n_ref=1000)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    dev = torch.device("cuda", engine.device)
    engine.set_groups(g)
    M = (sparse.csr_matrix if fmt == "csr" else sparse.csc_matrix)(X)
    d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (M.data, M.indices, M.indptr))
    want = oracle.run(M, g, is_log1p=True)
    got = engine.run_sparse(fmt, d, i, p, M.shape, 0, m, is_log1p=True)
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"f64 holding f32 values, is_log1p, {fmt}")
    engine.set_option("no_f64_narrowing", 1)
    try:
        wide = engine.run_sparse(fmt, d, i, p, M.shape, 0, m, is_log1p=True)
    finally:
        engine.set_option("no_f64_narrowing", 0)
    for a, b in zip(got, wide):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("fmt", ["csr", "csc"])
def test_host_sparse_count_values_travel_as_bytes(engine, fmt):
    """Host-resident sparse input whose stored values are counts below 255: the values go up as bytes (a quarter of the float32 array's
    share of the link) and are widened on the device -- the same planes, bit for bit, as with the array uploaded as it is, and the
    oracle's; a single value of 300 somewhere behind the sampled look makes the attempt stop and the array go up in its own type."""
    rng = np.random.RandomState(31)
    n, m = 60_000, 700
    X = (rng.poisson(rng.uniform(0.5, 12.0, size=m), size=(n, m)) * (rng.rand(n, m) < 0.11)).astype(np.float32)
    labels = make_labels(rng, n, 25, n_ref=3000)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    engine.set_groups(g)
    M = (sparse.csr_matrix if fmt == "csr" else sparse.csc_matrix)(X)
    assert M.nnz > 4 << 20
    run = lambda Mx: engine.run_sparse(fmt, Mx.data, Mx.indices, Mx.indptr, Mx.shape, 0, m)
    b0 = engine.input_bytes()
    got = run(M)
    as_bytes = engine.input_bytes() - b0
    engine.set_option("no_sparse_byte_values", 1)
    try:
        b0 = engine.input_bytes()
        plain = run(M)
        as_values = engine.input_bytes() - b0
    finally:
        engine.set_option("no_sparse_byte_values", 0)
    for a, b in zip(got, plain):
        np.testing.assert_array_equal(a, b)
    assert as_values - as_bytes == 3 * M.nnz, (as_values, as_bytes, M.nnz)       # one byte instead of four per stored value
    cols = [0, 233, 466, 699]
    want = oracle.run(np.ascontiguousarray(X[:, cols]), g)
    assert_planes_match(tuple(a[:, cols] for a in got), want, ref_row=g.encoded_ref_group, what=f"host {fmt}, values as bytes")
    M2 = M.copy()
    M2.data[M2.nnz // 2 + 12345] = 300.0
    X2 = M2.toarray()
    b0 = engine.input_bytes()
    got2 = run(M2)
    assert engine.input_bytes() - b0 >= as_values                                # (whatever went up as bytes before the value was met, plus the array itself)
    assert_planes_match(tuple(a[:, cols] for a in got2), oracle.run(np.ascontiguousarray(X2[:, cols]), g), ref_row=g.encoded_ref_group,
                        what=f"host {fmt}, one value beyond a byte")


@pytest.mark.parametrize("fmt", ["csr", "csc"])
@pytest.mark.parametrize("test,values", [("ovo", "counts"), ("ovo", "continuous"), ("ovr", "continuous")])
def test_sparse_input_with_more_groups_than_the_regrouping_kernels_hold(engine, fmt, test, values):
    """45 000 groups (two or three cells each): beyond the LDS histogram of the sparse regrouping kernels (~40 000 groups), which round 4
    refused (NotImplementedError).  The reference has no such limit (ovr/sparse_ovr.py:23-97, utils/groups.py:18-58): count-valued CSR
    keeps its group-major pass, everything else is written out as a dense window in the matrix's own type and takes the dense routes."""
    import torch
    rng = np.random.RandomState(45)
    G, n, m = 45_000, 110_000, 48
    codes = np.concatenate([np.zeros(2_000, dtype=np.int64), 1 + rng.randint(0, G - 1, size=n - 2_000 - (G - 1)), np.arange(1, G)])
    rng.shuffle(codes)
    labels = np.where(codes == 0, "non-targeting", np.char.add("pert_", np.char.zfill(codes.astype(str), 6)))
    X = rng.poisson(rng.uniform(0.5, 8.0, size=m), size=(n, m)).astype(np.float32)
    if values == "continuous":
        X = np.log1p(X * rng.uniform(0.5, 1.5, size=(n, m))).astype(np.float32)
    X *= rng.rand(n, m) < 0.15
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    assert g.counts.size == G
    engine.set_groups(g)
    M = (sparse.csr_matrix if fmt == "csr" else sparse.csc_matrix)(X)
    dev = torch.device("cuda", engine.device)
    d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (M.data, M.indices, M.indptr))
    got = engine.run_sparse(fmt, d, i, p, M.shape, 0, m)
    # (sparse OVR, a seventh of the cells stored: the dense routes' exact-integer tie sums against the reference's float64 ones -- 1e-12 of p)
    assert_planes_match(got, oracle.run(X, g, n_threads=4), ref_row=g.encoded_ref_group, what=f"45 000 groups {fmt} {test} {values}")
    got_h = engine.run_sparse(fmt, M.data, M.indices, M.indptr, M.shape, 8, 40)          # host arrays, a column window
    for a, b in zip(got_h, got):
        np.testing.assert_array_equal(a, b[:, 8:40])
