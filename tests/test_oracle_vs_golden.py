"""Pins the CPU oracle (oracle/) against outputs of the reference itself (tests/golden/*.npz,
produced by tests/golden/make_goldens.py) and against scipy.stats.mannwhitneyu, the reference's
own test oracle (reference tests/test_asymptotic_wilcoxon.py:63-108)."""
import numpy as np
import pytest
from scipy import sparse
from scipy.stats import mannwhitneyu, rankdata

import oracle
from conftest import assert_planes_match, load_golden, make_counts, make_labels

CASES = ["c1_1k_200_10", "small_ragged", "sparse90", "continuous"]


def _keys(z):
    return [k for k in z.files if k.count("|") == 4]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_goldens(name):
    z = load_golden(name)
    X, labels, ref = z["X"], z["labels"], str(z["reference"])
    mats = {"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}
    for key in _keys(z):
        fmt, test, alt, cc, tc = key.split("|")
        uniq, g = oracle.encode_and_count_groups(labels, ref if test == "ovo" else None)
        np.testing.assert_array_equal(uniq, z["groups"])
        got = oracle.run(mats[fmt], g, use_continuity=bool(int(cc)), tie_correct=bool(int(tc)), alternative=alt,
                         batch_size=16, n_threads=2)
        gold = z[key]
        want = (gold[:, :, 0], gold[:, :, 1], gold[:, :, 2])
        # dense OVO leaves the reference row uninitialised in the reference (dense_ovo.py:119-120)
        assert_planes_match(got, want, ref_row=g.encoded_ref_group, p_rtol=0.0, fc_rtol=0.0, what=f"{name} {key}")


@pytest.mark.parametrize("name", CASES)
def test_oracle_log1p_fold_change(name):
    z = load_golden(name)
    X, labels, ref = z["X"], z["labels"], str(z["reference"])
    _, g = oracle.encode_and_count_groups(labels, ref)
    got = oracle.run(np.log1p(X), g, is_log1p=True, batch_size=16)
    gold = z["dense|ovo|log1p"]
    # expm1 is taken in float32 (math.py:212); numpy's and glibc's expm1f may differ in the last f32 bit
    np.testing.assert_allclose(got[2], gold[:, :, 2], rtol=1e-6, equal_nan=True)
    mask = np.arange(got[0].shape[0]) != g.encoded_ref_group
    np.testing.assert_array_equal(got[1][mask], gold[mask][:, :, 1])


def test_primitives_match_reference():
    z = load_golden("primitives")
    for t in range(8):
        rs, ts = oracle.rank_sum_and_ties_from_sorted(z[f"merge{t}_A"], z[f"merge{t}_B"])
        np.testing.assert_array_equal([rs, ts], z[f"merge{t}_out"])
    for t in range(4):
        groups = z[f"acc{t}_groups"]
        rsums, ts = oracle.accumulate_group_ranksums_from_argsort(z[f"acc{t}_arr"], z[f"acc{t}_idx"], groups,
                                                                  z[f"acc{t}_ranksums"].size)
        np.testing.assert_array_equal(rsums, z[f"acc{t}_ranksums"])
        np.testing.assert_array_equal([ts], z[f"acc{t}_tie"])
    alts = {0: "two-sided", 1: "less", 2: "greater"}
    for n_ref, n_tgt, tie, U, cc, alt, pv in z["pval_rows"]:
        got = oracle.compute_pval(int(n_ref), int(n_tgt), int(n_ref + n_tgt), tie, U, n_ref * n_tgt / 2.0, cc, alts[int(alt)])
        np.testing.assert_allclose(got, pv, rtol=1e-15, atol=0.0)


def test_merge_rank_vs_rankdata():
    # mirrors reference tests/utils/test_ranking.py:13-32
    rng = np.random.RandomState(0)
    A = np.sort(rng.randint(0, 10, size=20)).astype(np.float64)
    B = np.sort(rng.randint(0, 10, size=15)).astype(np.float64)
    rs, ts = oracle.rank_sum_and_ties_from_sorted(A, B)
    comb = np.concatenate([A, B])
    assert rs == rankdata(comb)[len(A):].sum()
    _, c = np.unique(comb, return_counts=True)
    assert ts == (c ** 3 - c).sum()


@pytest.mark.parametrize("fmt", ["dense", "csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("alternative", ["two-sided", "less", "greater"])
@pytest.mark.parametrize("use_continuity", [True, False])
def test_oracle_vs_scipy(fmt, test, alternative, use_continuity):
    # fixture recipe of reference tests/conftest.py:76-100, smaller
    X, rng = make_counts(0, 2000, 15, 0.5)
    labels = make_labels(rng, 2000, 5)
    ref = labels[0] if test == "ovo" else None
    uniq, g = oracle.encode_and_count_groups(labels, ref)
    M = {"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}[fmt]
    p, u, fc = oracle.run(M, g, use_continuity=use_continuity, alternative=alternative, batch_size=16)
    for k, lab in enumerate(uniq):
        if lab == ref:
            continue
        grp = X[labels == lab]
        rest = X[labels == ref] if ref is not None else X[labels != lab]
        st, pv = mannwhitneyu(rest, grp, axis=0, method="asymptotic", use_continuity=use_continuity,
                              alternative=alternative)
        np.testing.assert_array_equal(u[k], st)
        np.testing.assert_allclose(p[k], pv, rtol=1e-12, atol=0.0)
        np.testing.assert_allclose(fc[k], grp.mean(axis=0, dtype=np.float64) / rest.mean(axis=0, dtype=np.float64), rtol=1e-12)


def test_oracle_errors():
    X, rng = make_counts(0, 100, 8, 0.5)
    labels = make_labels(rng, 100, 3)
    with pytest.raises(ValueError):
        oracle.encode_and_count_groups(labels, "nope")
    _, g = oracle.encode_and_count_groups(labels, None)
    with pytest.raises(ValueError):
        oracle.run(X, g, col_lb=0, col_ub=9)
    with pytest.raises(ValueError):
        oracle.run(X, g, alternative="bigger")
    assert oracle.check_indices_sorted_per_parcel([0, 2, 1, 3], [0, 2, 4])
    assert not oracle.check_indices_sorted_per_parcel([2, 0, 1, 3], [0, 2, 4])


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_reference_behaviour_on_explicitly_stored_zeros_is_pinned(test):
    """The reference's sparse kernels take every STORED entry for a value above the column's implicit zeros (sparse_ovo.py:74,
    sparse_ovr.py:77); its dense kernels rank the same numbers as numbers.  tests/golden/stored_zeros.npz holds both outputs of
    the reference for one matrix with explicit zeros: the oracle reproduces each with the matching input format, bit for bit --
    and they differ in every statistic (the HIP engine returns the dense numbers for sparse input too: DESIGN.md section 1,
    tests/test_gpu_sparse.py)."""
    from scipy import sparse
    z = load_golden("stored_zeros")
    labels, ref = z["labels"], str(z["reference"])
    _, g = oracle.encode_and_count_groups(labels, ref if test == "ovo" else None)
    M = sparse.csc_matrix((z["csc_data"], z["csc_indices"], z["csc_indptr"]), shape=z["X"].shape)
    assert (M.data == 0).sum() > 50
    got_sparse = oracle.run(M, g)
    got_dense = oracle.run(z["Xz"], g)
    for got, key in ((got_sparse, f"csc|{test}"), (got_dense, f"dense|{test}")):
        gold = z[key]
        mask = np.ones(gold.shape[0], dtype=bool)
        if test == "ovo" and key.startswith("dense"):
            mask[g.encoded_ref_group] = False   # the reference leaves this row uninitialised in its dense path
        np.testing.assert_array_equal(got[1][mask], gold[mask][:, :, 1], err_msg=key)
        np.testing.assert_array_equal(got[0][mask], gold[mask][:, :, 0], err_msg=key)
        np.testing.assert_array_equal(got[2], gold[:, :, 2], err_msg=key)
    rows = np.arange(got_dense[1].shape[0]) != (g.encoded_ref_group if test == "ovo" else -1)
    assert (got_sparse[1][rows] != got_dense[1][rows]).all()
