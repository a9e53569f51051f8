"""GPU parity tests of the packed dense OVO route (illico_amd/csrc/kernels_ovo_compact.h): group-wise packing of the non-zero
keys (k_group_compact) + look-ups in the bucketed reference (k_ovo_rank_compact), tie-heavy genes handed to k_ovo_rank.

Replaces the reference's dense OVO kernel (illico/ovo/dense_ovo.py:65-137).  Bar: statistics (2U, tie sums) identical to
the transpose + sort route and to the oracle's planes; p-values / fold changes at rtol 1e-12.
"""
import numpy as np
import pytest

import oracle
from conftest import assert_planes_match, make_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "eq-buckets", "narrow-tiles"])
def engine(request):
    """"eq-buckets": the packed rank kernel's distribution-following bucket function (on its own only for references above
    16384 cells) forced on every case.  "narrow-tiles": k_group_compact's 32-gene tiles (on their own only with blocks of 8192 rows
    and more: cluster-sized groups) on every case."""
    from illico_amd._lib import get_engine
    eng = get_engine()
    eng.set_option("no_fused_path", 1)  # every gene through the two-pass routes
    eng.set_option("packed_eq_buckets", 1 if request.param == "eq-buckets" else -1)
    eng.set_option("compact_narrow_rows", 1 if request.param == "narrow-tiles" else 0)
    yield eng
    eng.set_option("compact_narrow_rows", 0)
    eng.set_option("no_fused_path", 0)
    eng.set_option("no_packed_dense", 0)
    eng.set_option("packed_eq_buckets", -1)
    eng.set_option("profile", 0)


def _both_routes(engine, X, g, lb=0, ub=None, **kw):
    """(planes, statistics, kernel profile) of the packed route and of the route it replaces, same engine, same input."""
    ub = X.shape[1] if ub is None else ub
    engine.set_groups(g)
    out = []
    for off in (0, 1):
        engine.set_option("no_packed_dense", off)
        engine.set_option("profile", 1)
        engine.profile_reset()
        try:
            planes = engine.run_dense(X, lb, ub, **kw)
            prof = engine.profile_get()
        finally:
            engine.set_option("profile", 0)
        stats = engine.rank_statistics(X, lb, ub, is_log1p=kw.get("is_log1p", False))
        out.append((planes, stats, prof))
    engine.set_option("no_packed_dense", 0)
    return out


def _check(engine, X, labels, *, what, fc_rtol=1e-12, **kw):
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    (p_new, s_new, prof_new), (p_old, s_old, prof_old) = _both_routes(engine, X, g, **kw)
    assert "k_group_compact" in prof_new and "k_ovo_rank_compact" in prof_new and "k_transpose_permute" not in prof_new, prof_new
    assert "k_group_compact" not in prof_old and "k_transpose_permute" in prof_old, prof_old
    np.testing.assert_array_equal(s_new[0], s_old[0], err_msg=f"2U {what}")
    np.testing.assert_array_equal(s_new[1], s_old[1], err_msg=f"tie sums {what}")
    np.testing.assert_allclose(s_new[2], s_old[2], rtol=1e-13, atol=0, err_msg=f"value sums {what}")
    lb, ub = kw.get("lb", 0), kw.get("ub", X.shape[1])
    Xh = X.cpu().numpy() if hasattr(X, "cpu") else X
    want = oracle.run(np.ascontiguousarray(Xh, dtype=np.float64), g, col_lb=lb, col_ub=ub, **{k: v for k, v in kw.items() if k not in ("lb", "ub")})
    assert_planes_match(p_new, want, ref_row=g.encoded_ref_group, fc_rtol=fc_rtol, what=what)
    return prof_new


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_packed_route_matches_sort_route_and_oracle(engine, dtype):
    """Continuous columns, a tie-heavy column, a column of few distinct values, mixed signs, a constant and an all-zero
    column, ragged groups (1 .. ~90 cells), a reference of three packing segments."""
    rng = np.random.RandomState(5)
    n, m = 6000, 70
    labels = make_labels(rng, n, 90, n_ref=1300)
    X = np.log1p(rng.poisson(4.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m))) * (rng.rand(n, m) < 0.5)
    X[:, 3] = rng.poisson(300.0, size=n)                      # large counts: ties everywhere -> k_ovo_rank
    X[:, 4] = rng.choice([0.0, 0.25, 0.5, 7.75], size=n)      # four distinct values
    X[:, 5] = rng.randn(n)                                    # mixed sign, no zeros
    X[:, 6] = 2.5                                             # constant
    X[:, 7] = 0.0
    X[:, 8] = -np.abs(X[:, 8])                                # negative with zeros
    X[:, 9] = np.where(rng.rand(n) < 0.02, 3.25, 0.0)         # almost empty
    if np.issubdtype(dtype, np.integer):
        X = np.round(X * 1000.0)
    X = X.astype(dtype)
    prof = _check(engine, X, labels, what=f"packed {np.dtype(dtype).name}")
    assert "k_ovo_rank" in prof  # the tie-heavy genes (and only those workgroups do work)


def test_packed_route_duplicates_across_rounds_and_with_the_reference(engine):
    """A group of 150 cells (three rounds of 64 keys) whose keys repeat across rounds, repeat a reference key, and repeat a
    reference key that itself repeats: the Bloom flags + ballot comparison must give the exact multiplicities."""
    rng = np.random.RandomState(11)
    n, m = 3000, 8
    labels = np.array(["non-targeting"] * 700 + ["pert_a"] * 150 + ["pert_b"] * 40 + [f"pert_{i % 30:03d}" for i in range(n - 890)])
    X = rng.uniform(0.1, 5.0, size=(n, m)).astype(np.float32) * (rng.rand(n, m) < 0.9)
    a0 = 700
    X[a0 + 140, 0] = X[a0 + 3, 0]                # same group, rounds 2 and 0
    X[a0 + 70, 0] = X[a0 + 3, 0]                 # ... and round 1: a triple
    X[a0 + 10, 1] = X[5, 1]                      # ties with one reference key
    X[a0 + 100, 1] = X[5, 1]                     # ... twice
    X[6, 2] = X[7, 2] = X[8, 2]                  # a repeated reference key
    X[a0 + 20, 2] = X[6, 2]; X[a0 + 149, 2] = X[6, 2]
    X[a0:a0 + 150, 3] = np.float32(1.5)          # a group of one value
    perm = rng.permutation(n)
    prof = _check(engine, X[perm], labels[perm], what="duplicates")
    assert "k_ovo_rank" in prof


def test_packed_route_groups_beyond_256_nonzeros_go_to_k_ovo_rank(engine):
    rng = np.random.RandomState(12)
    n, m = 4000, 6
    labels = np.array(["non-targeting"] * 600 + ["pert_big"] * 700 + [f"pert_{i % 50:03d}" for i in range(n - 1300)])
    X = rng.uniform(0.1, 5.0, size=(n, m)).astype(np.float32)
    X[:, :3] *= rng.rand(n, 3) < 0.2             # genes 0-2: the big group keeps <= 256 non-zeros (about 140)
    perm = rng.permutation(n)
    _check(engine, X[perm], labels[perm], what="big group")


@pytest.mark.parametrize("lb,ub", [(0, 37), (1, 36), (3, 4), (5, 70)])
def test_packed_route_column_windows_and_unaligned_rows(engine, lb, ub):
    """Windows that start off a 16-byte boundary and a row pitch that is not a multiple of 4 elements (scalar loads)."""
    import torch
    rng = np.random.RandomState(13)
    n, m = 2500, 70
    labels = make_labels(rng, n, 25, n_ref=300)
    X = (rng.gamma(2.0, 1.0, size=(n, m)) * (rng.rand(n, m) < 0.4)).astype(np.float32)
    _check(engine, X, labels, what=f"window {lb}:{ub}", lb=lb, ub=ub)
    Xd = torch.zeros((n, m + 3), dtype=torch.float32, device="cuda")[:, :m]  # pitch 73
    Xd.copy_(torch.from_numpy(X))
    _check(engine, Xd, labels, what=f"device window {lb}:{ub}", lb=lb, ub=ub)


def test_packed_route_log1p_fold_change_and_options(engine):
    rng = np.random.RandomState(14)
    n, m = 2000, 20
    labels = make_labels(rng, n, 12, n_ref=250)
    X = np.log1p(rng.poisson(3.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m))).astype(np.float32)
    # expm1 is evaluated in float32 (utils/math.py:212): device and libm expm1f may differ by an f32 ulp
    _check(engine, X, labels, what="log1p", fc_rtol=1e-6, is_log1p=True)
    _check(engine, X, labels, what="less", alternative="less", use_continuity=False, tie_correct=False)


def test_packed_route_tiny_cases(engine):
    rng = np.random.RandomState(15)
    for n_ref, others in [(1, [1, 1, 2]), (2, [1]), (65, [64, 65, 63]), (513, [3])]:
        labels = np.array(["non-targeting"] * n_ref + sum(([f"pert_{i}"] * k for i, k in enumerate(others)), []))
        X = rng.uniform(-1, 1, size=(labels.size, 5)).astype(np.float32) * (rng.rand(labels.size, 5) < 0.7)
        _check(engine, X, labels, what=f"tiny {n_ref} {others}")


def test_packed_routes_many_tiny_groups_and_exact_round_sizes(engine):
    """Blocks of hundreds of 1 - 3 cell groups (a 64-key piece of a block then spans dozens of groups: the group codes of the OVR
    partition, the per-group offsets of the OVO rank kernel), and groups of exactly 64 / 65 / 128 / 192 / 256 / 257 non-zero keys
    (the rank kernel's round boundaries; 257 leaves for k_ovo_rank)."""
    rng = np.random.RandomState(18)
    tiny = np.concatenate([np.full(rng.randint(1, 4), i) for i in range(1500)])
    labels = np.array(["non-targeting"] * 300 + [f"tiny_{c:04d}" for c in tiny] +
                      sum(([f"exact_{k:03d}"] * k for k in (64, 65, 128, 192, 256, 257)), []))
    n = labels.size
    X = rng.uniform(0.1, 9.0, size=(n, 6)).astype(np.float32)   # no zeros: a group's non-zeros = its cells
    X[:, 1] *= rng.rand(n) < 0.5
    X[:, 2] = np.round(X[:, 2], 1)
    perm = rng.permutation(n)
    X, labels = X[perm], labels[perm]
    _check(engine, X, labels, what="tiny groups ovo")
    _check_ovr(engine, X, labels, what="tiny groups ovr")


def test_packed_route_reference_larger_than_its_key_slots(engine):
    """A reference of 30 000 cells does not fit the packed kernel's LDS key slots cell for cell: the slots are sized for non-zero
    keys (about 19 900 beside 2^17 buckets), a gene with more non-zeros than that is left to k_ovo_rank on the device, a sparser one is
    ranked by the packed kernel."""
    rng = np.random.RandomState(17)
    n_ref, others = 30000, 24
    labels = np.array(["non-targeting"] * n_ref + [f"pert_{i % others:02d}" for i in range(others * 90)])
    n = labels.size
    X = rng.gamma(2.0, 1.0, size=(n, 4)).astype(np.float32)
    X[:, 0] *= rng.rand(n) < 0.2          # 6 000 reference non-zeros: fits
    X[:, 2] *= rng.rand(n) < 0.72         # ~21 600: more than the slots
    X[:, 3] = np.round(X[:, 3], 1) * (rng.rand(n) < 0.5)  # ties, 15 000 non-zeros
    perm = rng.permutation(n)
    _check(engine, X[perm], labels[perm], what="large reference")


def test_reference_beyond_65535_cells_keeps_the_packed_route(engine):
    """(Until round 4's tenth sweep such a reference sent every gene to the general sort route: the route asked for a reference k_ovo_rank's
    LDS holds, which only its leftover genes need.)  Host-resident input; 30 % of the cells stored: 20 000 reference keys per gene."""
    rng = np.random.RandomState(16)
    n = 70000
    labels = np.array(["non-targeting"] * 66000 + [f"pert_{i % 20}" for i in range(n - 66000)])
    X = (rng.uniform(0.1, 3.0, size=(n, 3)) * (rng.rand(n, 3) < 0.3)).astype(np.float32)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    engine.set_groups(g)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = engine.run_dense(X, 0, 3)
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
    assert "k_ovo_rank_compact" in prof, prof
    assert_planes_match(got, oracle.run(X.astype(np.float64), g), ref_row=g.encoded_ref_group, what="large reference")


# ---- dense OVR: k_group_compact in place of the transposition, group sums folded in; the partition walks the packed rows (non-zero
# keys only) or, with "no_ovr_packed_partition", the padded rows (every key) -------------------------------------------------
def _check_ovr(engine, X, labels, *, what, fc_rtol=1e-12, **kw):
    for padded_rows in (0, 1):
        engine.set_option("no_ovr_packed_partition", padded_rows)
        try:
            _check_ovr_once(engine, X, labels, what=f"{what} padded_rows={padded_rows}", fc_rtol=fc_rtol, **kw)
        finally:
            engine.set_option("no_ovr_packed_partition", 0)


def _check_ovr_once(engine, X, labels, *, what, fc_rtol=1e-12, **kw):
    _, g = oracle.encode_and_count_groups(labels, None)
    (p_new, s_new, prof_new), (p_old, s_old, prof_old) = _both_routes(engine, X, g, **kw)
    assert "k_group_compact" in prof_new and "k_transpose_permute" not in prof_new, prof_new
    assert "k_group_compact" not in prof_old and "k_transpose_permute" in prof_old, prof_old
    np.testing.assert_array_equal(s_new[0], s_old[0], err_msg=f"2U {what}")
    np.testing.assert_array_equal(s_new[1], s_old[1], err_msg=f"tie sums {what}")
    np.testing.assert_allclose(s_new[2], s_old[2], rtol=1e-13, atol=0, err_msg=f"value sums {what}")
    lb, ub = kw.get("lb", 0), kw.get("ub", X.shape[1])
    Xh = X.cpu().numpy() if hasattr(X, "cpu") else X
    want = oracle.run(np.ascontiguousarray(Xh, dtype=np.float64), g, col_lb=lb, col_ub=ub, **{k: v for k, v in kw.items() if k not in ("lb", "ub")})
    assert_planes_match(p_new, want, fc_rtol=fc_rtol, what=what)


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64])
def test_padded_ovr_route_matches_transpose_route_and_oracle(engine, dtype):
    rng = np.random.RandomState(21)
    n, m = 5000, 70
    labels = make_labels(rng, n, 60)
    X = np.log1p(rng.poisson(4.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m))) * (rng.rand(n, m) < 0.5)
    X[:, 3] = rng.poisson(300.0, size=n)       # ties everywhere
    X[:, 5] = rng.randn(n)                     # mixed sign, no zeros
    X[:, 6] = 2.5
    X[:, 7] = 0.0
    if np.issubdtype(dtype, np.integer):
        X = np.round(X * 1000.0)
    _check_ovr(engine, X.astype(dtype), labels, what=f"padded ovr {np.dtype(dtype).name}")


@pytest.mark.parametrize("lb,ub", [(0, 37), (1, 36), (5, 70)])
def test_padded_ovr_route_windows_unaligned_rows_and_log1p(engine, lb, ub):
    import torch
    rng = np.random.RandomState(22)
    n, m = 3000, 70
    labels = np.array([f"pert_{i % 7}" for i in range(1500)] + ["pert_big"] * 1200 + [f"pert_small_{i % 150}" for i in range(300)])
    X = (rng.gamma(2.0, 1.0, size=(n, m)) * (rng.rand(n, m) < 0.4)).astype(np.float32)
    perm = rng.permutation(n)
    X, labels = X[perm], labels[perm]
    _check_ovr(engine, X, labels, what=f"ovr window {lb}:{ub}", lb=lb, ub=ub)
    Xd = torch.zeros((n, m + 3), dtype=torch.float32, device="cuda")[:, :m]  # pitch 73
    Xd.copy_(torch.from_numpy(X))
    _check_ovr(engine, Xd, labels, what=f"ovr device window {lb}:{ub}", lb=lb, ub=ub)
    # expm1 is evaluated in float32 (utils/math.py:212): device and libm expm1f may differ by an f32 ulp
    _check_ovr(engine, np.log1p(X), labels, what="ovr log1p", fc_rtol=1e-6, lb=lb, ub=ub, is_log1p=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_groups_of_thousands_of_cells_are_walked_in_sorted_pieces(engine, dtype):
    """Dense continuous OVO with groups above 1024 cells (clusters: the common non-perturbation use): each such group's packed run is
    sorted in LDS (k_sort_big_runs) and looked up in pieces of at most 256 keys cut at run boundaries.  Values rounded to three
    decimals put many duplicates inside a group -- across piece borders too --; a gene that holds one value in 400 cells of a big
    group (a run longer than a piece), a count-valued (tie-heavy) gene and a group too large for the LDS sort buffer all leave the
    route for the general one.  Statistics against the oracle and, bit for bit, against the transposition + sort route."""
    import torch
    rng = np.random.RandomState(2024)
    sizes = [3000, 2600, 1500, 1025, 700, 300, 257, 40, 1]   # group 0: the reference
    labels = np.concatenate([["non-targeting"] * sizes[0]] + [[f"c{i:02d}"] * sz for i, sz in enumerate(sizes[1:])])
    rng.shuffle(labels)
    n, m = labels.size, 48
    X = np.where(rng.rand(n, m) < 0.55, np.round(np.log1p(rng.poisson(4.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m))), 3), 0.0)
    X[:, 3] = np.where(rng.rand(n) < 0.6, np.round(rng.rand(n) * 3, 1), 0.0)                       # ~30 distinct values: long runs everywhere
    X[labels == "c00", 7] = np.where(rng.rand(2600) < 0.2, 1.234, X[labels == "c00", 7])           # one value ~500 times in a big group
    X[:, 11] = rng.poisson(2.0, size=n) * (rng.rand(n) < 0.5)                                       # counts: tie-heavy reference column
    X[:, 12] = -X[:, 12]                                                                            # negatives
    X[:, 13] = 0.0                                                                                  # an empty gene
    X = X.astype(dtype)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X.astype(np.float64) if dtype == np.float64 else X, g)
    Xd = torch.from_numpy(X).cuda()
    (p_new, s_new, prof_new), (p_old, s_old, prof_old) = _both_routes(engine, Xd, g)
    assert "k_group_compact" in prof_new and "k_ovo_rank_compact" in prof_new, prof_new
    assert "k_ovr_gene" in prof_new, prof_new                                                       # ... and the genes that left took the general route
    np.testing.assert_array_equal(s_new[0], s_old[0], err_msg="2U")
    np.testing.assert_array_equal(s_new[1], s_old[1], err_msg="tie sums")
    assert_planes_match(p_new, want, ref_row=g.encoded_ref_group, what=f"big groups, packed {np.dtype(dtype).name}")
    assert_planes_match(p_old, want, ref_row=g.encoded_ref_group, what=f"big groups, general {np.dtype(dtype).name}")
    got = engine.run_dense(X, 5, 40)                                                                # host input, a column window
    assert_planes_match(got, oracle.run(X, g, col_lb=5, col_ub=40), ref_row=g.encoded_ref_group, what="big groups, host window")


@pytest.mark.parametrize("density", [0.08, 0.6])
def test_dense_ovo_continuous_with_a_reference_of_more_than_65535_cells(engine, density):
    """The control group of an atlas: 70 000 reference cells, continuous values.  The packed rank kernel keeps as many of the reference's
    NON-ZERO keys as LDS holds: at 8 % density (log-normalised counts) every gene fits and takes it; at 60 % none does and every gene
    goes on to the general sort route -- both against the oracle.  (The route used to be closed to references above 65535 cells.)"""
    import torch
    rng = np.random.RandomState(321)
    n, m = 84_000, 70
    labels = np.array(["non-targeting"] * 70_000 + [f"p{i % 30:02d}" for i in range(n - 70_000)])
    rng.shuffle(labels)
    X = (np.exp(rng.normal(0.0, 1.0, size=(n, m))) * (rng.rand(n, m) < density)).astype(np.float32)
    X[:, 3] = 0
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X.astype(np.float64), g)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_dense(torch.from_numpy(X).to(torch.device("cuda", engine.device)), 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_ovo_rank_compact" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what=f"ovo continuous, reference of 70 000 cells, density {density}")


def test_dense_ovo_continuous_with_a_ranked_group_of_more_than_65535_cells(engine):
    """Cluster against cluster on continuous values: a ranked group of 70 000 cells.  A tenth of the values stored: the group's run of 7000
    keys per gene is dealt into value buckets and ranked piece by piece (packed rank kernel); a gene stored in full -- a run of 70 000 keys,
    beyond the 16-bit run length (its exact length travels beside it) and the bucket kernel's LDS slots -- is dealt through HBM and ranked by
    the same kernel; the column with ties (value buckets above 256 keys) is redone by the general route."""
    import torch
    rng = np.random.RandomState(654)
    n, m = 88_000, 66
    labels = np.array(["ref"] * 5_000 + ["big"] * 70_000 + [f"p{i % 25:02d}" for i in range(n - 75_000)])
    rng.shuffle(labels)
    X = (np.exp(rng.normal(0.0, 1.0, size=(n, m))) * (rng.rand(n, m) < 0.1)).astype(np.float32)
    X[:, 4] = np.exp(rng.normal(0.0, 1.0, size=n)).astype(np.float32)      # stored in full
    X[:, 9] = np.round(X[:, 9], 1)                                          # ties
    _, g = oracle.encode_and_count_groups(labels, "ref")
    want = oracle.run(X.astype(np.float64), g)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_dense(torch.from_numpy(X).to(torch.device("cuda", engine.device)), 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_ovo_rank_compact" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what="ovo continuous, a ranked group of 70 000 cells")


def test_dense_ovo_continuous_with_a_reference_of_600000_cells(engine):
    """More than 1024 reference segments of 512 rows (the rank kernel counts the reference's non-zeros segment by segment: once one thread
    per segment, with 1024 threads)."""
    import torch
    rng = np.random.RandomState(987)
    n, m = 700_000, 24
    labels = np.array(["ref"] * 600_000 + [f"p{i % 20:02d}" for i in range(n - 600_000)])
    rng.shuffle(labels)
    X = (np.exp(rng.normal(0.0, 1.0, size=(n, m))) * (rng.rand(n, m) < 0.02)).astype(np.float32)
    X[:, 2] = np.round(X[:, 2], 1)
    _, g = oracle.encode_and_count_groups(labels, "ref")
    want = oracle.run(X.astype(np.float64), g)
    engine.set_groups(g)
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_dense(torch.from_numpy(X).to(torch.device("cuda", engine.device)), 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_ovo_rank_compact" in prof, prof
    assert_planes_match(got, want, ref_row=g.encoded_ref_group, what="ovo continuous, reference of 600 000 cells")


@pytest.mark.parametrize("density", [0.1, 0.85])
def test_dense_ovr_continuous_with_clusters_of_more_than_65535_cells(engine, density):
    """rank_genes_groups on an atlas, one-versus-rest: clusters of 70 000 and 9 000 cells beside small groups, continuous values.  The
    partition walks the PACKED rows (non-zero keys only) whatever the group sizes: a group of 65535 cells and more is the last of its block,
    whose saturated 16-bit length is never looked at (its end is the block's key count); long blocks are dealt over all the wavefronts
    in 512-key units (k_ovr_partition_packed, `coop`), and the units' group codes start from a ballot over the group ends.  At 85 % density
    the big group's run per gene is 59 500 keys.  Against the oracle and, bit for bit, against the padded-rows form."""
    import torch
    rng = np.random.RandomState(77)
    n, m = 92_000, 40
    labels = np.array(["big"] * 70_000 + ["mid"] * 9_000 + [f"p{i % 40:02d}" for i in range(n - 79_000)])
    rng.shuffle(labels)
    X = (np.exp(rng.normal(0.0, 1.0, size=(n, m))) * (rng.rand(n, m) < density)).astype(np.float32)
    X[:, 5] = np.round(X[:, 5], 1)                                          # ties
    X[:, 6] = 0
    X[:, 7] = rng.randn(n)                                                  # no zeros, both signs
    _, g = oracle.encode_and_count_groups(labels, None)
    want = oracle.run(X.astype(np.float64), g)
    engine.set_groups(g)
    Xd = torch.from_numpy(X).to(torch.device("cuda", engine.device))
    engine.profile(True)
    engine.profile_reset()
    got = engine.run_dense(Xd, 0, m)
    prof = engine.profile_get()
    engine.profile(False)
    assert "k_group_compact" in prof and "k_ovr_partition" in prof, prof
    assert_planes_match(got, want, what=f"ovr continuous, clusters of 70 000 cells, density {density}")
    stats = engine.rank_statistics(Xd, 0, m)
    for opt in ("no_ovr_packed_big", "no_ovr_part_coop"):
        engine.set_option(opt, 1)
        try:
            old = engine.run_dense(Xd, 0, m)
            stats_old = engine.rank_statistics(Xd, 0, m)
        finally:
            engine.set_option(opt, 0)
        np.testing.assert_array_equal(got[0], old[0], err_msg=opt)
        np.testing.assert_array_equal(got[1], old[1], err_msg=opt)
        np.testing.assert_array_equal(stats[0], stats_old[0], err_msg=opt)
        np.testing.assert_array_equal(stats[1], stats_old[1], err_msg=opt)
