"""Primitive-level device checks, in the spirit of the reference's tests/utils/test_ranking.py:13-56: the rank sum and tie
sum of ONE column, read through illico_rank_statistics before any finalisation, against the reference's own primitive
outputs (tests/golden/primitives.npz: rank_sum_and_ties_from_sorted, ranking.py:52-158, and
_accumulate_group_ranksums_from_argsort, ranking.py:7-49) and against scipy.stats.rankdata."""
import numpy as np
import pytest
from scipy.stats import rankdata

import oracle
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from illico_amd._lib import get_engine
    return get_engine()


@pytest.fixture(params=["default", "sort-route", "sorted-reference"])
def route(request, engine):
    opts = {"no_counts_path": 0, "no_ovo_ref_buckets": 0, "no_ovr_parts_path": 0}
    if request.param != "default":
        opts["no_counts_path"] = 1          # integers too through the sort / bucket routes
    if request.param == "sorted-reference":
        opts.update(no_ovo_ref_buckets=1, no_ovr_parts_path=1)
    for k, v in opts.items():
        engine.set_option(k, v)
    yield request.param
    for k in opts:
        engine.set_option(k, 0)


def _ovo_stats(engine, A, B, dtype):
    """(rank sum of B in the merged sample, tie sum) from the device, for reference values A and group values B."""
    col = np.concatenate([A, B]).astype(dtype)[:, None]
    labels = np.array(["a"] * len(A) + ["b"] * len(B))
    perm = np.random.RandomState(len(A) * 131 + len(B)).permutation(len(labels))   # cell order must not matter
    _, g = oracle.encode_and_count_groups(labels[perm], "a")
    engine.set_groups(g)
    two_u, tie, _ = engine.rank_statistics(np.ascontiguousarray(col[perm]), 0, 1)
    nA, nB = len(A), len(B)
    return nA * nB + nB * (nB + 1) / 2 - two_u[0, 1] / 2, float(tie[0, 1])


def _ovr_stats(engine, arr, groups, n_groups, dtype):
    g = oracle.GroupContainer(groups.astype(np.int64), np.bincount(groups, minlength=n_groups).astype(np.int64),
                              np.argsort(groups, kind="stable").astype(np.int64),
                              np.concatenate([[0], np.cumsum(np.bincount(groups, minlength=n_groups))]).astype(np.int64), -1)
    engine.set_groups(g)
    two_u, tie, _ = engine.rank_statistics(np.ascontiguousarray(arr.astype(dtype)[:, None]), 0, 1)
    n, n_g = len(arr), g.counts.astype(np.float64)
    return (n - n_g) * n_g + n_g * (n_g + 1) / 2 - two_u[0] / 2, tie[0].astype(np.float64)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_rank_sum_and_ties_from_sorted_goldens(engine, route, dtype):
    z = load_golden("primitives")
    done = 0
    for t in range(8):
        A, B = z[f"merge{t}_A"], z[f"merge{t}_B"]
        if A.size == 0:
            continue  # an empty reference group cannot be expressed through a GroupContainer (groups.py:40-41)
        rs, ts = _ovo_stats(engine, A, B, dtype)
        np.testing.assert_array_equal([rs, ts], z[f"merge{t}_out"], err_msg=f"merge{t}")
        done += 1
    assert done >= 6


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_accumulate_group_ranksums_goldens(engine, route, dtype):
    z = load_golden("primitives")
    for t in range(4):
        want = z[f"acc{t}_ranksums"]
        rs, tie = _ovr_stats(engine, z[f"acc{t}_arr"], z[f"acc{t}_groups"], want.size, dtype)
        np.testing.assert_array_equal(rs, want, err_msg=f"acc{t}")
        np.testing.assert_array_equal(tie, np.full(want.size, z[f"acc{t}_tie"][0]), err_msg=f"acc{t} tie")


@pytest.mark.parametrize("nA,nB", [(20, 15), (1, 1), (64, 63), (65, 129), (300, 7), (1000, 257), (5000, 1025)])
@pytest.mark.parametrize("kind", ["ties", "continuous"])
def test_merge_rank_vs_rankdata(engine, route, nA, nB, kind):
    """reference tests/utils/test_ranking.py:13-32 at sizes around the wavefront / chunk boundaries."""
    rng = np.random.RandomState(nA * 7 + nB)
    if kind == "ties":
        A, B = rng.randint(0, 10, size=nA).astype(np.float64), rng.randint(0, 10, size=nB).astype(np.float64)
    else:
        A, B = rng.randn(nA), rng.randn(nB)
        k = min(nA, nB // 3)
        B[:k] = A[:k]   # some cross ties
    for dtype in (np.float32, np.float64):
        Ad, Bd = A.astype(dtype).astype(np.float64), B.astype(dtype).astype(np.float64)
        rs, ts = _ovo_stats(engine, Ad, Bd, dtype)
        comb = np.concatenate([Ad, Bd])
        assert rs == rankdata(comb)[nA:].sum()
        _, c = np.unique(comb, return_counts=True)
        assert ts == float((c.astype(np.int64) ** 3 - c).sum())


@pytest.mark.parametrize("n,G", [(30, 3), (64, 5), (1000, 17), (20000, 40)])
@pytest.mark.parametrize("kind", ["ties", "continuous", "half-zero"])
def test_group_ranksums_vs_rankdata(engine, route, n, G, kind):
    """reference tests/utils/test_ranking.py:35-56."""
    rng = np.random.RandomState(n + G)
    arr = rng.randint(0, 6, size=n).astype(np.float64) if kind == "ties" else rng.randn(n)
    if kind == "half-zero":
        arr[rng.rand(n) < 0.5] = 0.0
    groups = rng.randint(0, G, size=n)
    groups[:G] = np.arange(G)  # no empty group
    for dtype in (np.float32, np.float64):
        a = arr.astype(dtype).astype(np.float64)
        rs, tie = _ovr_stats(engine, a, groups, G, dtype)
        r = rankdata(a)
        want = np.array([r[groups == k].sum() for k in range(G)])
        np.testing.assert_array_equal(rs, want)
        _, c = np.unique(a, return_counts=True)
        np.testing.assert_array_equal(tie, np.full(G, float((c.astype(np.int64) ** 3 - c).sum())))
