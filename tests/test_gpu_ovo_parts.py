"""Continuous-value OVO beyond every LDS-resident look-up (illico_amd/csrc/kernels_ovo_compact.h):

* a reference whose non-zero keys outgrow the packed rank kernel's slots is taken in VALUE-RANGE PARTS (k_ovo_rank_compact<.., PARTS>:
  4096 cells over the reference's key range, cut at j / P of the keys; every part adds its share of S2 and of the tie terms);
* a ranked group's run of more keys than k_bucket_big_runs holds in LDS is dealt into value buckets through HBM.

Both replace the reference's linear merge for sizes of any kind (illico/utils/ranking.py:52-158, illico/ovo/dense_ovo.py:111-132).
The limits are lowered by options ("packed_ref_cap", "big_runs_cap") so that small matrices take the routes; every case is held to the
oracle (U exact, p rtol 1e-12) and, bit for bit, to the statistics of the same engine with the routes switched off.
"""
import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match, make_labels

pytestmark = pytest.mark.gpu


@pytest.fixture()
def engine():
    from illico_amd._lib import get_engine
    eng = get_engine()
    eng.set_option("no_fused_path", 1)
    yield eng
    for k in ("no_fused_path", "packed_ref_cap", "big_runs_cap", "no_ovo_parts", "no_big_runs_global", "no_deal_runs", "profile"):
        eng.set_option(k, 0)


def _continuous(rng, n, m, zero_frac):
    X = np.log1p(rng.poisson(6.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m)))
    X *= rng.rand(n, m) >= zero_frac
    return X


def _same(a, b):
    """Two routes of one engine: p-values and statistics bit for bit (integer statistics); fold changes to 1e-13 (dense value sums
    are fixed-order sums of each route's own order)."""
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_allclose(a[2], b[2], rtol=1e-13, atol=0)


def _profiled(engine, run):
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        planes = run()
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
    return planes, prof


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("zero_frac,cap", [(0.0, 1024), (0.4, 1024), (0.0, 2048), (0.4, 3072)])
def test_reference_in_value_range_parts(engine, dtype, zero_frac, cap):
    """A reference of 6000 cells against key slots capped at 1024 / 2048 / 3072: seven parts per gene (every part looks every key up,
    the other parts' lanes masked), four, or two (the groups' runs dealt by part first: k_deal_runs).  Columns: continuous, mixed sign without
    zeros (scaled data), ties between the reference and the groups and inside groups (values rounded to a grid), a column whose
    reference is one value (a cell of the cut holds everything: the gene leaves the route, the general route computes it), a constant
    column, one outlier stretching the key range."""
    rng = np.random.RandomState(12)
    n, m, G = 14000, 40, 30
    labels = make_labels(rng, n, G, n_ref=6000)
    X = _continuous(rng, n, m, zero_frac)
    X[:, 1] = rng.randn(n) * 0.7 + 1.0                        # mixed sign, no zeros
    X[:, 2] = np.round(rng.randn(n) + 1.0, 1)                 # ~80 distinct values of either sign: ties everywhere
    X[:, 3] = np.round(X[:, 3] * 8) / 8                       # a grid: ties among non-zeros
    X[:, 4] = np.where(labels == "non-targeting", 1.5, X[:, 4])   # the reference is ONE value
    X[:, 5] = 2.25
    X[7, 6] = 3.0e30                                          # an outlier: 4095 of the 4096 cells hold nothing
    X[:, 7] = -np.abs(X[:, 7]) - 0.5                          # all negative
    X = X.astype(dtype)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    engine.set_groups(g)
    want = oracle.run(np.ascontiguousarray(X, dtype=np.float64) if dtype == np.float64 else X, g)
    engine.set_option("packed_ref_cap", cap)
    got, prof = _profiled(engine, lambda: engine.run_dense(X, 0, m))
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"reference in parts {dtype.__name__} zeros {zero_frac} slots {cap}")
    engine.set_option("no_deal_runs", 1)
    try:
        _same(got, engine.run_dense(X, 0, m))                  # (dealt runs or masked look-ups: the same integers)
    finally:
        engine.set_option("no_deal_runs", 0)
    assert "k_ovo_rank_compact" in prof and "k_group_compact" in prof, prof
    stats = engine.rank_statistics(X, 0, m)
    engine.set_option("no_ovo_parts", 1)                      # (with the slots still capped: every gene leaves the kernel)
    old, prof_old = _profiled(engine, lambda: engine.run_dense(X, 0, m))
    stats_old = engine.rank_statistics(X, 0, m)
    _same(got, old)
    np.testing.assert_array_equal(stats[0], stats_old[0])
    np.testing.assert_array_equal(stats[1], stats_old[1])
    # without the parts every gene goes through the general routes (with them: the one-valued reference of gene 4 and what else crowds a part)
    left_old = prof_old.get("k_ovr_gene", {"launches": 0})["launches"] + prof_old.get("k_ovo_rank", {"launches": 0})["launches"]
    assert left_old > 0, (prof, prof_old)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_big_runs_dealt_through_hbm_and_parts_together(engine, dtype):
    """Groups of ~4000 cells (runs of ~2400 non-zero keys) against LDS slots capped at 512 keys: every run is dealt into value buckets
    through the second key buffer; the reference (5000 cells) is taken in parts at the same time (pieces wholly outside a part are
    skipped).  With ties inside the runs and a run that is ONE value (a bucket above 256 keys: that gene leaves the route)."""
    rng = np.random.RandomState(3)
    n, m, G = 25000, 24, 6
    labels = make_labels(rng, n, G, n_ref=5000)
    X = _continuous(rng, n, m, 0.4)
    X[:, 2] = np.round(X[:, 2] * 16) / 16
    X[:, 3] = rng.randn(n) * 0.7 + 1.0                        # mixed sign (a mean away from zero: group sums that cancel have no relative accuracy to hold)
    X[:, 4] = np.where(labels == "pert_00001", 0.75, X[:, 4])
    X = X.astype(dtype)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    engine.set_groups(g)
    want = oracle.run(X, g)
    engine.set_option("big_runs_cap", 512)
    got, prof = _profiled(engine, lambda: engine.run_dense(X, 0, m))
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"big runs through HBM {dtype.__name__}")
    engine.set_option("packed_ref_cap", 1024)
    got2, prof2 = _profiled(engine, lambda: engine.run_dense(X, 0, m))
    _same(got, got2)
    engine.set_option("no_big_runs_global", 1)
    engine.set_option("no_ovo_parts", 1)
    old, prof_old = _profiled(engine, lambda: engine.run_dense(X, 0, m))
    _same(got, old)
    # (gene 4 leaves the packed route either way; with the routes off, every gene does)
    t = lambda p: p.get("k_ovr_gene", {"launches": 0})["launches"] + p.get("k_ovo_rank", {"launches": 0})["launches"]
    assert "k_ovo_rank_compact" in prof and "k_ovo_rank_compact" in prof2 and t(prof_old) > 0, (prof, prof2, prof_old)


@pytest.mark.parametrize("fmt", ["csc", "csr"])
def test_sparse_input_with_cluster_sized_groups_parts_and_long_runs(engine, fmt):
    """The regrouped sparse route (k_csc_regroup / k_csc_segment + k_seg_to_packed + the packed rank kernel): five clusters of ~6000
    cells, a third of the cells stored -- reference runs of ~2000 keys against slots capped at 1024 (parts), ranked runs of ~2000 keys
    against LDS slots capped at 512 (through HBM)."""
    import torch
    rng = np.random.RandomState(8)
    n, m, G = 30000, 20, 5
    labels = np.array(["non-targeting"] * 6000 + [f"pert_{1 + i % (G - 1):05d}" for i in range(n - 6000)])
    rng.shuffle(labels)
    X = _continuous(rng, n, m, 0.67).astype(np.float32)
    X[:, 1] = np.round(X[:, 1] * 4) / 4
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    engine.set_groups(g)
    want = oracle.run(X, g)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    dev = torch.device("cuda", engine.device)
    d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (M.data, M.indices, M.indptr))
    run = lambda: engine.run_sparse(fmt, d, i, p, M.shape, 0, m)
    base = run()
    assert_planes_match(base, want, ref_row=g.encoded_ref_group, what=f"sparse clusters {fmt}")
    engine.set_option("packed_ref_cap", 1024)
    engine.set_option("big_runs_cap", 512)
    got, prof = _profiled(engine, run)
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"sparse clusters {fmt}, parts + runs through HBM")
    for a, b in zip(got, base):
        np.testing.assert_array_equal(a, b)
    assert "k_ovo_rank_compact" in prof, prof
