"""GPU parity tests (dense input): HIP engine through the C-ABI vs the CPU oracle and the golden vectors.

Bar (north star / reference tests/test_asymptotic_wilcoxon.py:166-185): statistic bit-exact,
p-value rtol 1e-12 atol 0, fold change rtol 1e-12 (is_log1p=False).
"""
import numpy as np
import pandas as pd
import pytest

import oracle
from conftest import assert_planes_match, load_golden, make_counts, make_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from illico_amd._lib import get_engine
    return get_engine()


_DEVICE_INPUT = False  # set by the `route` fixture: device-resident X takes the fused single-pass OVO route


def _run(engine, X, grpc, **kw):
    engine.set_groups(grpc)
    if _DEVICE_INPUT and isinstance(X, np.ndarray) and X.dtype in (np.float32, np.float64, np.int32, np.int64):
        import torch
        Xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()
        out = engine.run_dense(Xd, kw.pop("col_lb", 0), kw.pop("col_ub", X.shape[1]), **kw)
        engine.synchronize()
        np.testing.assert_array_equal(Xd.cpu().numpy(), X)  # input never mutated
        return out
    return engine.run_dense(X, kw.pop("col_lb", 0), kw.pop("col_ub", X.shape[1]), **kw)


@pytest.fixture(params=["fused+counts+sort", "fused-host+counts+sort", "fused-host-narrow+counts+sort", "packed", "counts+sort", "sort-only"])
def route(request, engine):
    """Dense OVO has four device routes per gene: the fused single-pass histogram kernel (integer values < 64; X
    device-resident, or a host matrix copied up in column windows), the packed route (group-wise packing + look-ups), and
    behind it the transpose route with the two-pass histogram kernel (integer values < 2048) and the general sort kernel.
    The params switch routes off so that each one is exercised on the same data."""
    global _DEVICE_INPUT
    _DEVICE_INPUT = request.param == "fused+counts+sort"
    engine.set_option("no_fused_path", 0 if request.param.startswith("fused") else 1)
    engine.set_option("no_packed_dense", 1 if request.param in ("counts+sort", "sort-only") else 0)
    engine.set_option("no_counts_path", 1 if request.param == "sort-only" else 0)
    # fused OVR has a one-pass form (per-group histograms) and a two-pass form: the host-input param runs the latter
    engine.set_option("no_ovr_one_pass", 1 if request.param.startswith("fused-host+") else 0)
    # a host matrix goes up as float32 windows ("fused-host": the byte form forbidden) or as byte windows ("fused-host-narrow": forced,
    # whatever the values: genes with a cell the bytes cannot hold come back and go up again in their own type)
    engine.set_option("host_narrow", 1 if request.param.startswith("fused-host-narrow") else (-1 if request.param.startswith("fused-host+") else 0))
    # ... and is itself the fallback of the value-range parts route (k_ovr_partition + k_csc_ovr_gene), which the fused
    # params leave on
    engine.set_option("no_ovr_parts_path", 0 if request.param.startswith("fused") else 1)
    # the OVO sort route keeps the reference column in value buckets (no sort) unless its values crowd; "sort-only" sorts it
    engine.set_option("no_ovo_ref_buckets", 1 if request.param == "sort-only" else 0)
    yield request.param
    engine.set_option("no_ovo_ref_buckets", 0)
    engine.set_option("no_ovr_parts_path", 0)
    engine.set_option("no_counts_path", 0)
    engine.set_option("no_fused_path", 0)
    engine.set_option("no_ovr_one_pass", 0)
    engine.set_option("no_packed_dense", 0)
    engine.set_option("host_narrow", 0)
    _DEVICE_INPUT = False


@pytest.mark.parametrize("name", ["c1_1k_200_10", "small_ragged", "sparse90", "continuous"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_dense_matches_reference_goldens(engine, name, test, route):
    z = load_golden(name)
    X, labels, ref = z["X"], z["labels"], str(z["reference"])
    from illico_amd.utils.groups import encode_and_count_groups
    for key in [k for k in z.files if k.startswith(f"dense|{test}|") and k.count("|") == 4]:
        _, _, alt, cc, tc = key.split("|")
        _, g = encode_and_count_groups(labels, ref if test == "ovo" else None)
        got = _run(engine, X, g, use_continuity=bool(int(cc)), tie_correct=bool(int(tc)), alternative=alt)
        gold = z[key]
        assert_planes_match(got, (gold[:, :, 0], gold[:, :, 1], gold[:, :, 2]), ref_row=g.encoded_ref_group,
                            what=f"{name} {key}")
        if test == "ovo":
            assert np.all(got[0][g.encoded_ref_group] == 1.0) and np.all(got[1][g.encoded_ref_group] == -1.0)


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_dense_dtypes_vs_oracle(engine, test, dtype, route):
    X, rng = make_counts(11, 3000, 70, 0.5)
    labels = make_labels(rng, 3000, 9, n_ref=300)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    Xd = X.astype(dtype)
    got = _run(engine, Xd, g)
    want = oracle.run(X.astype(np.float64), g)
    assert_planes_match(got, want, what=f"{test} {dtype}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_dense_negative_and_continuous_values(engine, test, route):
    rng = np.random.RandomState(5)
    X = rng.randn(2500, 33).astype(np.float32)
    X[rng.rand(*X.shape) < 0.3] = 0.0
    X[rng.rand(*X.shape) < 0.01] = -0.0
    labels = make_labels(rng, 2500, 8, n_ref=200)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    got = _run(engine, X, g)
    want = oracle.run(X, g)
    # mixed-sign sums cancel: fold change is compared at 1e-9 here, U and p at the usual bar
    assert_planes_match(got, want, fc_rtol=1e-9, what=test)


def test_ovo_ragged_group_sizes(engine, route):
    """Group sizes straddling every wave-sort width (1, 63..65, 127..129, 255..257, 511..513, 1023, 1024)."""
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 700]
    rng = np.random.RandomState(3)
    codes = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)])
    rng.shuffle(codes)
    labels = np.array([f"g{c:02d}" for c in codes])
    n = codes.size
    X = rng.poisson(rng.uniform(0.1, 8, size=21), size=(n, 21)).astype(np.float32)
    X[rng.rand(n, 21) < 0.4] = 0
    X[:, 3] = 0.0           # constant gene: tie_corr == 0 -> p = 1 (math.py:96,117-118)
    X[:, 4] = rng.rand(n)   # no ties at all
    for ref in ("g16", "g00", "g09"):
        _, g = oracle.encode_and_count_groups(labels, ref)
        got = _run(engine, X, g)
        want = oracle.run(X, g)
        assert_planes_match(got, want, what=f"ref={ref}")


@pytest.mark.parametrize("device_input", [False, True])
def test_ovo_counts_route_big_groups_and_mixed_genes(engine, device_input):
    global _DEVICE_INPUT
    _DEVICE_INPUT = device_input
    try:
        _counts_route_body(engine)
    finally:
        _DEVICE_INPUT = False


def _counts_route_body(engine):
    """Histogram route: group and reference sizes beyond what the sort route holds in registers / LDS, genes
    that must fall back to the sort route (fractional, negative, >= 2048) next to count-valued ones."""
    rng = np.random.RandomState(13)
    sizes = [45000, 3000, 1500, 700, 64, 1]  # reference 45000 cells (> 160 KiB of LDS keys), groups > 1024
    codes = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)])
    rng.shuffle(codes)
    labels = np.array([f"g{c:02d}" for c in codes])
    n = codes.size
    X = rng.poisson(rng.uniform(0.1, 30, size=9), size=(n, 9)).astype(np.float32)
    X[rng.rand(n, 9) < 0.5] = 0
    X[:, 5] = rng.randint(0, 2048, size=n)   # widest value range the table holds
    X[:, 6] = 0.0
    _, g = oracle.encode_and_count_groups(labels, "g00")
    want = oracle.run(X, g)
    got = _run(engine, X, g)
    assert_planes_match(got, want, what="counts route, big groups")
    for dt in (np.int32, np.float64, np.int64):
        assert_planes_match(_run(engine, X.astype(dt), g), want, what=f"counts route {dt}")
    # genes outside the tables take the sort route; at these sizes that is the global radix-sort fallback
    Xmix = X.copy()
    Xmix[0, 2] = 0.5
    Xmix[:, 7] = rng.randn(n)                     # continuous, mixed sign
    Xmix[:, 8] = np.round(rng.randn(n) * 2) / 2   # heavy ties at half-integers
    assert_planes_match(_run(engine, Xmix, g), oracle.run(Xmix, g), fc_rtol=1e-9, what="global-sort fallback")
    assert_planes_match(_run(engine, Xmix.astype(np.float64), g), oracle.run(Xmix.astype(np.float64), g), fc_rtol=1e-9,
                        what="global-sort fallback f64")
    # moderate sizes: mixed routes in one call
    sizes = [900, 700, 300, 64, 1]
    codes = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)])
    rng.shuffle(codes)
    labels = np.array([f"g{c:02d}" for c in codes])
    n = codes.size
    X = rng.poisson(3.0, size=(n, 12)).astype(np.float32)
    X[:, 1] += 0.25 * (rng.rand(n) < 0.1)   # fractional values
    X[:, 2] -= 1.0                          # negative values
    X[:, 3] = rng.randint(0, 5000, size=n)  # above the table
    X[3, 4] = 2048.0                        # exactly one value at the table limit
    _, g = oracle.encode_and_count_groups(labels, "g01")
    assert_planes_match(_run(engine, X, g), oracle.run(X, g), what="mixed routes")


def test_ovr_ragged(engine):
    rng = np.random.RandomState(4)
    sizes = [1, 5, 64, 1000, 3000, 129]
    codes = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)])
    rng.shuffle(codes)
    labels = np.array([f"g{c:02d}" for c in codes])
    n = codes.size
    X = rng.poisson(2.0, size=(n, 19)).astype(np.float32)
    X[:, 0] = 0.0
    X[:, 1] = rng.rand(n)
    X[:, 2] = 7.0
    _, g = oracle.encode_and_count_groups(labels, None)
    got = _run(engine, X, g)
    want = oracle.run(X, g)
    assert_planes_match(got, want, what="ovr ragged")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("sorted_form,parts_cap", [(0, 0), (1, 0), (0, 1024)])
def test_ovr_dense_value_range_parts_route(engine, dtype, sorted_form, parts_cap):
    """Dense OVR, any values: each gene's non-zero keys are split by value into parts that fit LDS and ranked part by
    part (k_ovr_partition + k_csc_ovr_gene<PARTS>; bucket form, and the sorted form when forced).  70 000 cells: fully
    dense columns need 3+ parts (5+ for float64 keys); half-empty and nearly empty columns, negatives, exact repeats
    among continuous values, a tie-heavy column and a constant column (crowded coarse buckets: parts of their own, counted per
    group when they hold one value; two values in one crowded bucket: that gene leaves the route and the general route recomputes
    it), an all-zero column.  parts_cap =
    1024 makes ~90 parts per dense column (more than 64: all 8 bits of the part id in play)."""
    rng = np.random.RandomState(509)
    n, m = 70000, 14
    sizes = [30000, 20000, 9000, 700, 300, 255, 40, 3, 1]
    sizes.append(n - sum(sizes))
    labels = np.concatenate([[f"s{i:02d}"] * sz for i, sz in enumerate(sizes)])
    rng.shuffle(labels)
    X = np.zeros((n, m))
    X[:, 0] = np.log1p(rng.poisson(3.0, size=n) * rng.uniform(0.5, 1.5, size=n))      # ~5 % exact zeros, rest spread
    X[:, 1] = rng.randn(n)                                                            # fully dense, negatives
    X[:, 2] = np.where(rng.rand(n) < 0.5, 0.0, rng.lognormal(0.0, 1.0, size=n))       # half empty
    X[:, 3] = np.where(rng.rand(n) < 0.999, 0.0, rng.rand(n))                         # nearly empty
    X[:, 4] = rng.poisson(2.0, size=n)                                                # tie-heavy: crowded buckets, one value each
    X[:, 5] = rng.rand(n) * 1e-3 + 5.0                                                # narrow range: one coarse bucket
    X[:, 6] = 0.0                                                                     # all zero
    X[:, 7] = np.round(rng.lognormal(0.0, 1.0, size=n), 2)                            # many exact repeats, wide range
    X[:, 8] = rng.exponential(1.0, size=n) * (rng.rand(n) < 0.9)
    X[:, 9] = 3.25                                                                    # constant
    X[:, 10] = rng.standard_cauchy(size=n)                                            # heavy tails both ways
    X[:, 11] = np.where(rng.rand(n) < 0.3, -rng.rand(n), rng.rand(n) * 100)
    # crowded coarse buckets (each gets a part of its own, beyond the key slots: the streaming form when it is ONE value, else the
    # general route): log1p of raw counts -- 64 distinct values, none an integer -- and two neighbouring values with an outlier that
    # stretches the key range so that both fall into one bucket
    X[:, 12] = np.log1p(rng.poisson(6.0, size=n) * (rng.rand(n) < 0.8))
    X[:, 13] = np.where(rng.rand(n) < 0.5, 1.0, np.nextafter(dtype(1.0), dtype(2.0)))
    X[7, 13] = 1.0e6
    X = X.astype(dtype)
    _, g = oracle.encode_and_count_groups(labels, None)
    want = oracle.run(X.astype(np.float64), g)
    engine.set_option("no_fused_path", 1)
    engine.set_option("csc_ovr_sorted_form", sorted_form)
    engine.set_option("ovr_parts_cap", parts_cap)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = _run(engine, X, g)
        prof = engine.profile_get()
        got_w = _run(engine, X, g, col_lb=1, col_ub=4, alternative="less", tie_correct=False)
    finally:
        engine.set_option("profile", 0)
        engine.set_option("csc_ovr_sorted_form", 0)
        engine.set_option("ovr_parts_cap", 0)
        engine.set_option("no_fused_path", 0)
    assert "k_ovr_partition" in prof and "k_ovr_rank_parts" in prof and "k_ovr_gene" in prof, prof
    assert_planes_match(got, want, what=f"dense ovr parts {dtype.__name__} sorted_form={sorted_form}")
    want_w = oracle.run(X.astype(np.float64), g, col_lb=1, col_ub=4, alternative="less", tie_correct=False)
    assert_planes_match(got_w, want_w, what=f"dense ovr parts window {dtype.__name__}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_column_window_batching_and_strides(engine, test, route):
    X, rng = make_counts(21, 2000, 301, 0.5)
    labels = make_labels(rng, 2000, 6, n_ref=150)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(X, g, col_lb=37, col_ub=290)
    engine.set_option("gene_batch", 64)
    try:
        got = _run(engine, X, g, col_lb=37, col_ub=290)
    finally:
        engine.set_option("gene_batch", 0)
    assert_planes_match(got, want, what="window")
    # output planes that are column windows of larger arrays (row stride != width)
    big = np.full((3, g.counts.size, 400), np.nan)
    out = tuple(big[k][:, 100:353] for k in range(3))
    _run(engine, X, g, col_lb=37, col_ub=290, out=out)
    assert_planes_match(out, want, what="strided out")
    assert np.isnan(big[:, :, :100]).all() and np.isnan(big[:, :, 353:]).all()
    with pytest.raises(ValueError):
        _run(engine, X, g, col_lb=5, col_ub=302)
    with pytest.raises(ValueError):
        _run(engine, X, g, alternative="bigger")


def test_device_resident_input_and_output(engine):
    import torch
    X, rng = make_counts(8, 4000, 130, 0.5)
    labels = make_labels(rng, 4000, 12, n_ref=300)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X, g)
    Xd = torch.from_numpy(X).cuda()
    engine.set_groups(g)
    outs = engine.run_dense(Xd, 0, X.shape[1], device_out=True)
    engine.synchronize()
    got = tuple(o.cpu().numpy() for o in outs)
    assert_planes_match(got, want, what="device io")
    np.testing.assert_array_equal(Xd.cpu().numpy(), X)  # input not mutated


def test_log1p_fold_change(engine):
    X, rng = make_counts(9, 1500, 40, 0.5)
    labels = make_labels(rng, 1500, 6, n_ref=120)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    Xl = np.log1p(X)
    got = _run(engine, Xl, g, is_log1p=True)
    want = oracle.run(Xl, g, is_log1p=True)
    # expm1 is evaluated in float32 (utils/math.py:212): device and libm expm1f may differ by an f32 ulp
    assert_planes_match(got, want, fc_rtol=1e-6, what="log1p")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_drop_in_call_dataframe(test):
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    z = load_golden("c1_1k_200_10")
    X, labels, ref = z["X"], z["labels"], str(z["reference"])
    adata = AnnDataLite(X.copy(), obs=pd.DataFrame({"pert": labels}))
    df = asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", reference=ref if test == "ovo" else None,
                             batch_size=64 if test == "ovo" else "auto")
    assert list(df.columns) == ["p_value", "statistic", "fold_change"]
    assert df.index.names == ["pert", "feature"]
    gold = z[f"dense|{test}|two-sided|1|1"]
    G = gold.shape[0]
    got = df.values.reshape(G, X.shape[1], 3)
    ref_row = int(np.flatnonzero(z["groups"] == ref)[0]) if test == "ovo" else None
    assert_planes_match((got[:, :, 0], got[:, :, 1], got[:, :, 2]), (gold[:, :, 0], gold[:, :, 1], gold[:, :, 2]),
                        ref_row=ref_row, what="dataframe")
    assert list(df.index.get_level_values(0).unique()) == list(z["groups"])
    np.testing.assert_array_equal(adata.X, X)  # input not mutated (reference tests :187-194)
    with pytest.raises(ValueError):
        asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", reference="nope")
    with pytest.raises(KeyError):
        asymptotic_wilcoxon(AnnDataLite([[1.0]], obs=pd.DataFrame({"pert": ["a"]})) if False else
                            type("A", (), {"X": object(), "layers": {}, "obs": {"pert": ["a"]}, "var_names": ["g"]})(),
                            is_log1p=False, group_keys="pert")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_backed_memmap_input_streams_chunks(tmp_path, test, monkeypatch):
    """Out-of-core dense input (np.memmap, the stand-in for the reference's h5py handler, registry.py:162-168):
    gene chunks are streamed with a prefetch thread; results equal the in-RAM call; the file is not modified."""
    import sys
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    aw = sys.modules["illico_amd.asymptotic_wilcoxon"]  # the module (the package attribute of that name is the function)
    X, rng = make_counts(77, 1500, 300, 0.5)
    X[:, 7] = rng.rand(1500)                       # a gene for the sort route
    labels = make_labels(rng, 1500, 7, n_ref=120)
    np.save(tmp_path / "x.npy", X)
    mm = np.load(tmp_path / "x.npy", mmap_mode="r")
    monkeypatch.setattr(aw, "STREAM_CHUNK_BYTES", 1500 * 4 * 37)   # 37 genes per chunk -> 9 chunks
    ref = "non-targeting" if test == "ovo" else None
    obs = pd.DataFrame({"pert": labels})
    got = asymptotic_wilcoxon(AnnDataLite(mm, obs=obs), is_log1p=False, group_keys="pert", reference=ref)
    want = asymptotic_wilcoxon(AnnDataLite(X, obs=obs), is_log1p=False, group_keys="pert", reference=ref)
    pd.testing.assert_frame_equal(got, want)
    np.testing.assert_array_equal(np.load(tmp_path / "x.npy"), X)


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_fused_route_second_pass_takes_counts_up_to_255(engine, test):
    """Counts of 64 .. 255 (highly expressed genes of a real count matrix) stay on the single-pass fused route: its second,
    256-value pass takes the genes the 64-value pass flags, on the device, without a host round trip.  Poisson(60) genes next
    to Poisson(3) genes; one gene with a count of 300 (beyond the wider table too) and one fractional gene still go to the
    two-pass routes.  Same bytes as with the second pass switched off; matches the oracle."""
    import torch
    rng = np.random.RandomState(67)
    n, m = 9000, 200
    means = np.where(np.arange(m) % 3 == 0, 60.0, 3.0)
    X = (rng.poisson(means, size=(n, m)) * (rng.rand(n, m) < 0.6)).astype(np.float32)
    X[17, 33] = 300.0
    X[:, 34] = np.where(rng.rand(n) < 0.1, 0.5, X[:, 34])
    X[:, 36] = 255.0 * (rng.rand(n) < 0.5)     # the last value of the wider table
    labels = make_labels(rng, n, 40, n_ref=800)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    rr = g.encoded_ref_group if test == "ovo" else None
    want = oracle.run(X, g)
    Xd = torch.from_numpy(X).cuda()
    engine.set_groups(g)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = engine.run_dense(Xd, 0, m)
        prof = engine.profile_get()
        engine.set_option("no_fused_wide", 1)
        engine.profile_reset()
        narrow_only = engine.run_dense(Xd, 0, m)
        prof2 = engine.profile_get()
    finally:
        engine.set_option("no_fused_wide", 0)
        engine.set_option("profile", 0)
    assert "k_ovo_fused_wide" in prof and "k_ovo_fused_wide" not in prof2, (prof, prof2)
    # with the second pass only genes 33 and 34 are left for the two-pass routes, without it a third of the matrix; either way
    # the leftover genes of this count matrix are gathered into a narrow matrix and take the histogram routes
    assert "k_gather_columns" in prof and "k_gather_columns" in prof2, (prof, prof2)
    counts_kernel = "k_ovo_counts" if test == "ovo" else "k_ovr_counts"
    assert counts_kernel in prof2 and prof[counts_kernel]["ms"] < prof2[counts_kernel]["ms"], (prof, prof2)
    for a, b in zip(got, narrow_only):
        assert a.tobytes() == b.tobytes()
    assert_planes_match(got, want, ref_row=rr, what=f"counts up to 255 {test}")
    # the same through CSR byte windows (k_csr_densify writes bytes: values up to 254 fit)
    from scipy import sparse
    M = sparse.csr_matrix(X[:, :120])
    got = engine.run_sparse("csr", M.data, M.indices, M.indptr, M.shape, 0, 120)
    assert_planes_match(got, tuple(a[:, :120] for a in want), ref_row=rr, what=f"csr byte window, counts up to 254 {test}")


@pytest.mark.parametrize("big_group", [False, True])
def test_ovo_counts_table_is_4096_values_while_groups_stay_below_256_cells(engine, big_group):
    """k_ovo_counts keeps 8-bit multiplicities and a 4096-value table while no ranked group exceeds 255 cells, 16-bit ones and 2048
    values otherwise: genes with counts in [2048, 4096) take the histogram kernel in the first case and the sort routes in the
    second; both match the oracle bit for bit (tie-heavy values around the limits included)."""
    rng = np.random.RandomState(91)
    n, m = 6000, 12
    sizes = [300] + ([400] if big_group else []) + [200] * 20
    labels = np.concatenate([np.full(k, "non-targeting" if i == 0 else f"pert_{i:02d}") for i, k in enumerate(sizes)])
    labels = np.concatenate([labels, [f"rest_{i % 30:02d}" for i in range(n - labels.size)]])
    rng.shuffle(labels)
    X = rng.poisson(3.0, size=(n, m)).astype(np.float32)
    X[:, 2] = rng.poisson(3000.0, size=n)                          # [2048, 4096)
    X[:, 3] = rng.randint(2040, 2056, size=n)                      # ties on both sides of 2048
    X[:, 4] = rng.randint(4090, 4100, size=n)                      # ... and of 4096: beyond the table
    X[:, 5] = np.where(rng.rand(n) < 0.7, 0, rng.randint(1, 4096, size=n))
    X[:, 6] = 4095.0
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X, g)
    engine.set_groups(g)
    engine.set_option("no_fused_path", 1)
    engine.set_option("no_packed_dense", 1)
    engine.profile(True)
    engine.profile_reset()
    try:
        got = engine.run_dense(X, 0, m)
    finally:
        engine.set_option("no_fused_path", 0)
        engine.set_option("no_packed_dense", 0)
        prof = engine.profile_get()
        engine.profile(False)
    assert "k_ovo_counts" in prof, prof
    assert ("k_ovo_rank" in prof or "k_ovr_gene" in prof), prof    # column 4 (and, with a big group, columns 2, 3, 5, 6 too)
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"4096-value table big_group={big_group}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("where", ["device", "device-deferred", "host"])
def test_scattered_big_count_genes_are_gathered_and_take_the_histogram_routes(engine, test, where):
    """A count matrix whose highly expressed genes lie beyond the fused tables (64 .. 255: second pass; beyond: the genes are
    gathered into a narrow matrix and take k_ovo_counts / k_ovr_counts, a fractional gene and one beyond every table the sort
    routes) -- device-resident, deferred and host-resident input (the host windows gather their flagged columns on the device).  Most
    tiles hold a flagged gene here, so on device-resident input the 256-value stage is left to the host and runs on the gathered
    columns (k_wide_decide); identical to recomputing the column runs (`no_leftover_gather`) and to the stage run in place
    (`no_wide_gather`)."""
    import torch
    rng = np.random.RandomState(77)
    n, m, G = 20000, 640, 40
    means = np.exp(rng.normal(2.0, 1.8, size=m)).clip(0.05, 3000.0)
    X = rng.poisson(means, size=(n, m)).astype(np.float32)
    X[rng.rand(n, m) < 0.5] = 0
    X[:, 100] = X[:, 100] * 0.5 + 0.25 * (X[:, 100] > 0)     # fractional values
    X[:50, 200] = 40000.0 + np.arange(50)                     # beyond k_ovr_counts' table as well
    mx = X.max(axis=0)
    assert (mx > 63).sum() > 60 and (mx > 255).sum() > 10 and (mx > 2047).sum() >= 2 and ((mx > 63).sum() < m // 2)
    labels = make_labels(rng, n, G, n_ref=1500)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(X, g)
    engine.set_groups(g)
    dev = torch.device("cuda", engine.device)

    def run():
        if where == "host":
            return engine.run_dense(X, 0, m)
        Xd = torch.from_numpy(X).to(dev)
        if where == "device":
            return engine.run_dense(Xd, 0, m)
        out = tuple(torch.full((g.counts.size, m), -7.0, dtype=torch.float64, device=dev) for _ in range(3))
        engine.run_dense(Xd, 0, m, out=out, defer=True)
        engine.synchronize()
        return tuple(t.cpu().numpy() for t in out)

    engine.profile(True)
    engine.profile_reset()
    got = run()
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_gather_columns" in prof and ("k_ovo_counts" if test == "ovo" else "k_ovr_counts") in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"big counts {test} {where}")
    for option in ("no_leftover_gather", "no_wide_gather"):  # ... and to running the 256-value stage over the window as it lies
        engine.set_option(option, 1)
        try:
            again = run()
        finally:
            engine.set_option(option, 0)
        for a, b in zip(got, again):
            np.testing.assert_array_equal(a, b, err_msg=option)


@pytest.mark.parametrize("where", ["device", "host"])
def test_dense_ovr_counts_with_a_group_of_more_than_65535_cells(engine, where):
    """OVR on a count matrix whose control group holds 70 000 cells (an atlas of two million cells has one): the one-pass form of the
    fused OVR route counts in 16-bit cells per group, the two-pass form keeps no per-group state -- it must take over, not the general
    sort route (115 ms instead of 5 at 2 000 000 x 1200).  Genes beyond the tables ride along."""
    import torch
    rng = np.random.RandomState(123)
    n, m = 90_000, 130
    labels = np.array(["ctrl"] * 70_000 + [f"p{i % 40:02d}" for i in range(n - 70_000)])
    rng.shuffle(labels)
    X = rng.poisson(rng.uniform(0.2, 12.0, size=m), size=(n, m)).astype(np.float32)
    X[rng.rand(n, m) < 0.4] = 0
    X[:, 7] = rng.poisson(90.0, size=n)        # beyond the 64-value table
    X[:, 11] = rng.poisson(400.0, size=n)      # beyond the 256-value table
    _, g = oracle.encode_and_count_groups(labels, None)
    assert g.counts.max() == 70_000
    want = oracle.run(X, g)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_dense(torch.from_numpy(X).to(torch.device("cuda", engine.device)) if where == "device" else X, 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert ("k_group_value_hists" if where == "device" else "k_ovr_fused") in prof or "k_group_value_hists" in prof, prof
    assert_planes_match(got, want, what=f"ovr, a group of 70 000 cells, {where}")


@pytest.mark.parametrize("where", ["device", "host"])
def test_dense_ovo_counts_with_a_ranked_group_of_more_than_65535_cells(engine, where):
    """Cluster against cluster: a ranked group of 70 000 cells.  The fused OVO pass keeps a running multiplicity per (group, value) -- 8 or
    16 bits -- and was closed to such groups (every gene then took the radix sort in HBM: 158 ms at one million cells x 2400 genes); with
    32-bit cells it takes them.  A gene beyond the tables goes its own way."""
    import torch
    rng = np.random.RandomState(77)
    n, m = 90_000, 130
    labels = np.array(["ref"] * 6_000 + ["big"] * 70_000 + [f"p{i % 20:02d}" for i in range(n - 76_000)])
    rng.shuffle(labels)
    X = rng.poisson(rng.uniform(0.2, 12.0, size=m), size=(n, m)).astype(np.float32)
    X[rng.rand(n, m) < 0.4] = 0
    X[:, 5] = 3.0                               # one value: a tie block of 70 000
    X[:, 7] = rng.poisson(90.0, size=n)         # beyond the 64-value table
    _, g = oracle.encode_and_count_groups(labels, "ref")
    want = oracle.run(X, g)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_dense(torch.from_numpy(X).to(torch.device("cuda", engine.device)) if where == "device" else X, 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_group_value_hists" in prof or "k_ovo_fused" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"ovo, a ranked group of 70 000 cells, {where}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32])
def test_few_large_groups_take_the_group_histogram_route(engine, test, dtype):
    """Count-valued dense input with few, large groups (clusters): the rows of a group are split over many wavefronts that count
    (group, gene) value histograms (k_group_value_hists), the statistics come from the histograms (k_emit_from_group_hists:
    kernels_group_hists.h).  Against the oracle and, bit for bit (planes and integer statistics), against the fused kernels the route
    replaces -- ragged groups (1 .. 9000 cells, an empty stretch boundary in the middle of a wavefront's positions), a one-valued column
    (a tie block of the whole column), an all-zero column, columns beyond the 64-value table and a non-integer column (they leave
    the route), a column window."""
    import torch
    rng = np.random.RandomState(5150)
    sizes = [9000, 7000, 4000, 2500, 1300, 700, 64, 33, 2, 1]
    labels = np.concatenate([[f"c{i:02d}"] * sz for i, sz in enumerate(sizes)])
    rng.shuffle(labels)
    n, m = labels.size, 150
    X = rng.poisson(rng.uniform(0.2, 14.0, size=m), size=(n, m)).astype(np.float64)
    X[rng.rand(n, m) < 0.5] = 0
    X[:, 3] = 5
    X[:, 4] = 0
    X[:, 9] = rng.poisson(80.0, size=n)        # beyond the 64-value table
    X[:, 10] = rng.poisson(500.0, size=n)      # beyond the 256-value table
    X[:, 11] = rng.rand(n)                     # no counts at all
    X[:, 12] = 63 * (rng.rand(n) < 0.3)        # the table's last value
    X = X.astype(dtype)
    _, g = oracle.encode_and_count_groups(labels, "c02" if test == "ovo" else None)
    want = oracle.run(np.ascontiguousarray(X, dtype=np.float64) if dtype != np.float32 else X, g, col_lb=3, col_ub=m)
    engine.set_groups(g)
    Xd = torch.from_numpy(X).to(torch.device("cuda", engine.device))
    engine.set_option("group_hist_min_cells", 1)
    try:
        engine.profile(True)
        engine.profile_reset()
        got = engine.run_dense(Xd, 3, m)
        prof = engine.profile_get()
        engine.profile(False)
        stats = engine.rank_statistics(Xd, 3, m)
        engine.set_option("no_group_hist_route", 1)
        old = engine.run_dense(Xd, 3, m)
        stats_old = engine.rank_statistics(Xd, 3, m)
    finally:
        engine.set_option("no_group_hist_route", 0)
        engine.set_option("group_hist_min_cells", 0)
        engine.profile(False)
    assert "k_group_value_hists" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group if test == "ovo" else None, what=f"group histograms {test} {np.dtype(dtype).name}")
    for a, b in zip(got, old):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(stats[0], stats_old[0])
    np.testing.assert_array_equal(stats[1], stats_old[1])
