"""Degenerate shapes and values through the C-ABI (dense + sparse, host + device input) vs the oracle."""
import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from illico_amd._lib import get_engine
    return get_engine()


def _all_inputs(X):
    import torch
    yield "dense-host", lambda e, lb, ub, **kw: e.run_dense(X, lb, ub, **kw)
    Xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    yield "dense-device", lambda e, lb, ub, **kw: e.run_dense(Xd, lb, ub, **kw)
    for fmt, ctor in (("csc", sparse.csc_matrix), ("csr", sparse.csr_matrix)):
        M = ctor(X)
        yield fmt, (lambda e, lb, ub, M=M, fmt=fmt, **kw: e.run_sparse(fmt, M.data, M.indices, M.indptr, M.shape, lb, ub, **kw))


def _check(engine, X, labels, ref, what, **kw):
    _, g = oracle.encode_and_count_groups(labels, ref)
    want = oracle.run(X, g, **kw)
    engine.set_groups(g)
    for name, fn in _all_inputs(X):
        got = fn(engine, 0, X.shape[1], **kw)
        assert_planes_match(got, want, ref_row=g.encoded_ref_group if ref is not None else None,
                            what=f"{what} [{name}]")


@pytest.mark.parametrize("ref", ["a", None])
def test_two_groups_one_gene(engine, ref):
    rng = np.random.RandomState(0)
    X = rng.poisson(2.0, size=(50, 1)).astype(np.float32)
    labels = np.array(["a"] * 20 + ["b"] * 30)
    _check(engine, X, labels, ref, "2 groups x 1 gene")


@pytest.mark.parametrize("ref", ["g0", None])
def test_single_cell_groups_and_reference_of_one_cell(engine, ref):
    rng = np.random.RandomState(1)
    labels = np.array(["g0"] + ["g1"] + ["g2"] * 5 + ["g3"] * 70)
    X = rng.poisson(1.5, size=(labels.size, 7)).astype(np.float32)
    _check(engine, X, labels, ref, "single-cell groups")


@pytest.mark.parametrize("ref", ["a", None])
def test_all_zero_and_constant_matrices(engine, ref):
    labels = np.array(["a"] * 10 + ["b"] * 12 + ["c"] * 3)
    for X in (np.zeros((25, 5), np.float32), np.full((25, 5), 3.0, np.float32)):
        # tie_corr == 0 -> p = 1 (math.py:96,117-118); all-zero columns give fold change inf (0/0 -> inf, math.py:192)
        _check(engine, X, labels, ref, "constant matrix")


def test_ovr_single_group(engine):
    # one group: "the rest" is empty (n_ref = 0): the reference divides by zero the same way (nan / inf planes)
    X = np.arange(12, dtype=np.float32).reshape(6, 2)
    labels = np.array(["only"] * 6)
    _, g = oracle.encode_and_count_groups(labels, None)
    with np.errstate(all="ignore"):
        want = oracle.run(X, g)
    engine.set_groups(g)
    got = engine.run_dense(X, 0, 2)
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(np.isnan(got[0]), np.isnan(want[0]))


def test_empty_window_and_many_options(engine):
    rng = np.random.RandomState(2)
    X = rng.poisson(3.0, size=(300, 9)).astype(np.float32)
    labels = np.array([f"g{i % 4}" for i in range(300)])
    _, g = oracle.encode_and_count_groups(labels, "g1")
    engine.set_groups(g)
    p, u, fc = engine.run_dense(X, 4, 4)   # empty chunk: three [G, 0] planes (asymptotic_wilcoxon.py:49: lb == ub allowed)
    assert p.shape == (4, 0) and u.shape == (4, 0) and fc.shape == (4, 0)
    for alt in ("two-sided", "less", "greater"):
        for cc in (True, False):
            for tc in (True, False):
                _check(engine, X, labels, "g1", f"{alt} {cc} {tc}", alternative=alt, use_continuity=cc, tie_correct=tc)
                _check(engine, X, labels, None, f"ovr {alt} {cc} {tc}", alternative=alt, use_continuity=cc, tie_correct=tc)


def test_large_values_and_table_boundaries(engine):
    """Values around the histogram-table limits (63/64 and 2047/2048) and beyond, negative zero, huge floats."""
    rng = np.random.RandomState(3)
    n = 900
    labels = np.array([f"g{i % 6}" for i in range(n)])
    X = np.zeros((n, 8), np.float32)
    X[:, 0] = rng.randint(0, 64, size=n)
    X[:, 1] = rng.randint(0, 65, size=n)
    X[:, 2] = rng.randint(0, 2048, size=n)
    X[:, 3] = rng.randint(0, 2049, size=n)
    X[:, 4] = rng.randint(0, 3, size=n) * 1e30
    X[:, 5] = np.where(rng.rand(n) < 0.5, -0.0, 0.0) + rng.randint(0, 2, size=n)
    X[:, 6] = rng.randint(-3, 4, size=n)
    X[:, 7] = rng.randint(0, 4, size=n) + 0.5
    _check(engine, X, labels, "g2", "table boundaries ovo")
    _check(engine, X, labels, None, "table boundaries ovr")


@pytest.mark.parametrize("ref", ["g00000", None])
def test_many_groups_beyond_lds_accumulators(engine, ref):
    """20 000 groups: the radix routes keep their per-group accumulators in HBM instead of LDS (non-count data;
    for OVO one group is large enough to need the global-sort fallback)."""
    rng = np.random.RandomState(5)
    G = 20000
    codes = np.concatenate([np.zeros(1500, dtype=int), np.repeat(np.arange(1, G), 2)])
    rng.shuffle(codes)
    labels = np.array([f"g{c:05d}" for c in codes])
    n = codes.size
    X = np.where(rng.rand(n, 3) < 0.4, 0, rng.rand(n, 3)).astype(np.float32)
    X[:, 1] = np.round(X[:, 1] * 4) / 4   # ties
    _, g = oracle.encode_and_count_groups(labels, ref)
    want = oracle.run(X, g)
    engine.set_groups(g)
    got = engine.run_dense(X, 0, 3)
    assert_planes_match(got, want, ref_row=g.encoded_ref_group if ref else None, what="20k groups dense")
    M = sparse.csc_matrix(X)
    got = engine.run_sparse("csc", M.data, M.indices, M.indptr, M.shape, 0, 3)
    assert_planes_match(got, want, ref_row=g.encoded_ref_group if ref else None, what="20k groups csc")


@pytest.mark.parametrize("ref", ["s017", "s300", None])
def test_every_group_size_through_the_chunk_cascade(engine, ref):
    """The fused kernels walk a group in chunks of 32 rows, then 16, 8 and one predicated chunk of up to 7: one group
    of every size from 1 to 75 plus a few large ones (one above 255 cells: 16-bit multiplicity cells) exercises every
    combination; 70 genes = one full 64-gene tile and a partial one."""
    rng = np.random.RandomState(5)
    sizes = list(range(1, 76)) + [130, 300, 257]
    labels = np.concatenate([[f"s{sz:03d}"] * sz for sz in sizes])
    rng.shuffle(labels)
    X = (rng.poisson(2.5, size=(labels.size, 70)) * (rng.rand(labels.size, 70) < 0.6)).astype(np.float32)
    X[:, 3] = 0                      # an all-zero gene
    X[:, 9] = rng.randint(0, 64, size=labels.size)   # the whole table range
    _check(engine, X, labels, ref, "chunk cascade")


def test_more_cells_per_test_than_the_64bit_products_hold(engine):
    """n (n-1)(n+1) and the t^3 tie terms are 64-bit integer products (utils/math.py:95): beyond 2^21 - 1 cells in one test the
    reference's int64 wraps silently.  OVO (n_ref + n_tgt cells per test) is refused by illico_set_groups; OVR is refused for DENSE input
    at the call and taken for SPARSE input (below: test_sparse_ovr_over_three_million_cells)."""
    n = (1 << 21) + 5
    labels = np.array(["a", "b", "c"])[np.arange(n) % 3]
    _, g_ovr = oracle.encode_and_count_groups(labels, None)
    engine.set_groups(g_ovr)
    with pytest.raises(NotImplementedError, match="2097151"):
        engine.run_dense(np.zeros((n, 2), dtype=np.float32), 0, 2)
    _, g_ovo = oracle.encode_and_count_groups(labels, "a")     # 0.7M + 0.7M cells per test: fine
    engine.set_groups(g_ovo)
    big = np.where(np.arange(n) < n - 10, "a", "b")             # reference of 2^21 - 5 cells + a group of 10: 2^21 + 5 per test
    _, g_big = oracle.encode_and_count_groups(big, "a")
    with pytest.raises(NotImplementedError, match="overflow 64-bit"):
        engine.set_groups(g_big)


@pytest.mark.parametrize("fmt", ["csc", "csr"])
def test_sparse_ovr_over_three_million_cells(engine, fmt):
    """Atlas-scale OVR on sparse input: 3 000 000 cells in every test.  The reference cannot be the yardstick here -- its int64
    n (n-1)(n+1) wraps (utils/math.py:95) --, so the expected planes are built from scipy's ranks with Python integers (exact tie sums
    and products) and the reference's own float64 formulas (utils/math.py:95-104, sparse_ovr.py:49,83): U exact, p at 1e-12."""
    import math
    from scipy import sparse, stats
    rng = np.random.RandomState(99)
    n, m = 3_000_000, 3
    codes = rng.randint(0, 4, size=n)
    codes[:37] = 4                                             # a small group that sits on top: a tiny p-value
    labels = np.array([f"g{c}" for c in codes])
    X = np.zeros((n, m), dtype=np.float32)
    for j in range(m):
        nz = rng.rand(n) < 0.05
        v = np.round(np.exp(rng.randn(n) * 0.7) + 0.05, 2) if j < 2 else 1.0 + rng.poisson(2.0, size=n)
        X[:, j] = np.where(nz, v, 0)
    X[:37, :] = 50.0 + rng.rand(37, m).round(2)
    _, g = oracle.encode_and_count_groups(labels, None)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    engine.set_groups(g)
    got = engine.run_sparse(fmt, M.data, M.indices, M.indptr, M.shape, 0, m)
    G = g.counts.size
    for j in range(m):
        col = X[:, j].astype(np.float64)
        ranks = stats.rankdata(col)
        vals, cnt = np.unique(col[col != 0], return_counts=True)
        n0 = float(n - int(cnt.sum()))
        tie = float(sum(int(t) ** 3 - int(t) for t in cnt))    # the non-zero blocks: exact
        tie += n0 * n0 * n0 - n0                                # sparse_ovr.py:83, n0 a float64 (:49)
        nnn = float(n * (n - 1) * (n + 1))                      # Python integers: no wrap
        for k in range(G):
            n_t = int(g.counts[k]); n_r = n - n_t
            R = float(ranks[g.encoded_groups == k].sum())
            U = n_r * n_t + n_t * (n_t + 1) / 2 - R
            assert got[1][k, j] == U, (fmt, j, k)
            tie_corr = 1.0 - tie / nnn
            sigma = math.sqrt(float(n_r * n_t * (n_r + n_t + 1)) / 12.0 * tie_corr)
            mu = n_r * n_t / 2.0
            Um = min(U, n_r * n_t - U)
            d = Um - mu
            z = (abs(d) + (1.0 if d > 0 else (-1.0 if d < 0 else 0.0)) * 0.5) / sigma
            p = math.erfc(z / math.sqrt(2.0))
            assert got[0][k, j] == pytest.approx(p, rel=1e-12, abs=0), (fmt, j, k, got[0][k, j], p)
