"""Run-to-run and order-of-arrival independence of the device results, boundary hygiene (threads, torch streams).

Fold changes come from per-(gene, group) value sums; the reference adds them in cell-index order (utils/math.py:27-39,
196-221).  The device meets a sparse column's values in an order that depends on timing, so the sums are formed EXACTLY
(integer limbs, kernels_sums.h) and rounded once: the bytes cannot depend on the order, and for float32 data they equal
``math.fsum`` of the group's values.
"""
import math
import threading

import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match, make_counts, make_labels

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from illico_amd._lib import get_engine
    return get_engine()


def _continuous(seed, n, m, sparsity, dtype=np.float32):
    X, rng = make_counts(seed, n, m, sparsity)
    X = np.log1p(X * rng.uniform(0.5, 1.5, size=X.shape)).astype(dtype)
    return X, rng


def _run_sparse(engine, M, g, **kw):
    engine.set_groups(g)
    return engine.run_sparse(M.format, M.data, M.indices, M.indptr, M.shape, 0, M.shape[1], **kw)


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("route_opts", [{}, {"no_csc_gene_path": 1, "no_csc_ovr_gene_path": 1}, {"no_csr_transpose_path": 1}],
                         ids=["single-kernel", "two-kernel", "csr-regroup"])
def test_sparse_runs_are_byte_identical_and_match_oracle(engine, fmt, test, dtype, route_opts):
    X, rng = _continuous(21, 6000, 48, 0.85, dtype)
    labels = make_labels(rng, 6000, 40, n_ref=500)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    for k, v in route_opts.items():
        engine.set_option(k, v)
    try:
        runs = [_run_sparse(engine, M, g) for _ in range(3)]
    finally:
        for k in route_opts:
            engine.set_option(k, 0)
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert a.tobytes() == b.tobytes(), "two runs on the same input differ"
    assert_planes_match(runs[0], oracle.run(X.astype(np.float64), g), ref_row=g.encoded_ref_group if test == "ovo" else None,
                        what=f"{fmt} {test} {np.dtype(dtype).name}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_csc_entry_order_does_not_change_a_byte(engine, test):
    """The same matrix with each column's stored entries in a different order: identical planes, bit for bit."""
    X, rng = _continuous(22, 5000, 40, 0.8)
    labels = make_labels(rng, 5000, 25, n_ref=400)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    M = sparse.csc_matrix(X)
    M2 = M.copy()
    for j in range(M.shape[1]):
        s, e = M.indptr[j], M.indptr[j + 1]
        perm = rng.permutation(e - s)
        M2.data[s:e] = M.data[s:e][perm]
        M2.indices[s:e] = M.indices[s:e][perm]
    a = _run_sparse(engine, M, g)
    b = _run_sparse(engine, M2, g)
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_sparse_value_sums_are_correctly_rounded_exact_sums(engine, fmt, test):
    """float32 values spanning < 2^59: every value is exact in the 84-bit fixed point, so the device's group sums are the
    correctly rounded exact sums -- math.fsum -- and the fold change built from them is reproduced bit for bit."""
    X, rng = _continuous(23, 4000, 24, 0.8)
    X[:, 3] *= np.float32(1e-6)        # small magnitudes
    X[:, 4] *= np.float32(3e7)         # large magnitudes
    X[5:50, 5] *= np.float32(1e5)      # a wide range inside one gene
    labels = make_labels(rng, 4000, 12, n_ref=300)
    ref = "non-targeting" if test == "ovo" else None
    uniq, g = oracle.encode_and_count_groups(labels, ref)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    engine.set_option("no_dense_window_path", 1)
    try:
        fc = _run_sparse(engine, M, g)[2]
    finally:
        engine.set_option("no_dense_window_path", 0)
    G, m = len(uniq), X.shape[1]
    Xd = X.astype(np.float64)
    sums = np.array([[math.fsum(Xd[g.encoded_groups == k, j]) for j in range(m)] for k in range(G)])
    counts = g.counts.astype(np.float64)[:, None]
    mu = sums / counts
    with np.errstate(divide="ignore", invalid="ignore"):
        if test == "ovo":
            mu_ref = np.broadcast_to(mu[g.encoded_ref_group], mu.shape)
        else:
            total = np.zeros(m)
            for k in range(G):  # group order, like group_agg_counts.sum(axis=0) (math.py:185)
                total = total + sums[k]
            mu_ref = (total[None, :] - sums) / (X.shape[0] - counts)
        want = np.where(mu_ref == 0, np.inf, mu / mu_ref)
    assert fc.tobytes() == want.tobytes()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_dense_ovr_parts_runs_are_byte_identical(engine, dtype):
    X, rng = _continuous(24, 30000, 12, 0.4, dtype)
    labels = make_labels(rng, 30000, 30)
    _, g = oracle.encode_and_count_groups(labels, None)
    engine.set_groups(g)
    engine.set_option("ovr_parts_cap", 4096)  # several parts per gene
    try:
        runs = [engine.run_dense(X, 0, X.shape[1]) for _ in range(3)]
    finally:
        engine.set_option("ovr_parts_cap", 0)
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert a.tobytes() == b.tobytes()
    assert_planes_match(runs[0], oracle.run(X.astype(np.float64), g), what=f"dense ovr parts {np.dtype(dtype).name}")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_largest_key_values_are_not_confused_with_padding(engine, test):
    """INT32_MAX / INT64_MAX map onto the largest unsigned key, which the bucket walks also use as padding."""
    rng = np.random.RandomState(7)
    n, m = 3000, 10
    X = rng.randint(0, 5, size=(n, m)).astype(np.int64)
    big32 = np.iinfo(np.int32).max
    X[rng.rand(n, m) < 0.02] = big32
    X[:, 3] = np.where(rng.rand(n) < 0.5, big32, rng.randint(-5, 5, size=n))   # many copies of the largest key
    X[:, 4] = np.where(rng.rand(n) < 0.1, big32 - 1, X[:, 4])
    labels = make_labels(rng, n, 8, n_ref=400)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(X.astype(np.float64), g)
    rr = g.encoded_ref_group if test == "ovo" else None
    engine.set_groups(g)
    for dt in (np.int32, np.int64):
        Xi = X.astype(dt)
        got = engine.run_dense(Xi, 0, m)
        assert_planes_match(got, want, ref_row=rr, what=f"dense {np.dtype(dt).name} with INT32_MAX")
        for fmt in ("csc", "csr"):
            M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(Xi)
            got = engine.run_sparse(fmt, M.data, M.indices, M.indptr, M.shape, 0, m)
            assert_planes_match(got, want, ref_row=rr, what=f"{fmt} {np.dtype(dt).name} with INT32_MAX")
    X64 = X.copy()
    X64[X64 == big32] = np.iinfo(np.int64).max
    want64 = oracle.run(X64, g)
    got = engine.run_dense(X64, 0, m)
    np.testing.assert_array_equal(got[1][np.arange(len(got[1])) != (rr if rr is not None else -1)],
                                  want64[1][np.arange(len(got[1])) != (rr if rr is not None else -1)])


def test_four_threads_share_one_context(engine):
    """The reference's driver calls one dispatcher from joblib threads (asymptotic_wilcoxon.py:236-241): concurrent calls on
    ONE context are serialised by the library and each gets its own, correct planes."""
    X, rng = make_counts(31, 5000, 96, 0.5)
    Xc, _ = _continuous(32, 5000, 96, 0.5)
    labels = make_labels(rng, 5000, 20, n_ref=400)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    engine.set_groups(g)
    want = {0: oracle.run(X, g), 1: oracle.run(Xc.astype(np.float64), g)}
    chunks = [(k, lb, lb + 24) for k in (0, 1) for lb in range(0, 96, 24)]
    results, errors = {}, []

    def work(tid):
        try:
            for i, (k, lb, ub) in enumerate(chunks):
                if i % 4 == tid:
                    results[(k, lb)] = engine.run_dense(X if k == 0 else Xc, lb, ub)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (k, lb), got in results.items():
        w = tuple(a[:, lb:lb + 24] for a in want[k])
        assert_planes_match(got, w, ref_row=g.encoded_ref_group, what=f"thread chunk {k} {lb}")


def test_engine_follows_torch_current_stream(engine):
    """X produced asynchronously on a torch side stream just before the call, planes consumed on it just after: the engine
    runs on torch's current stream, so no explicit synchronisation is needed."""
    import torch
    X, rng = make_counts(33, 20000, 256, 0.5)
    labels = make_labels(rng, 20000, 50, n_ref=700)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(X, g)
    engine.set_groups(g)
    base = torch.from_numpy(X).cuda()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        Y = base
        for _ in range(40):            # a chain of async kernels the engine must wait for
            Y = Y + 1.0
        Y = Y - 40.0
        p, u, fc = engine.run_dense(Y, 0, 256, device_out=True)
        p2, u2, fc2 = p.clone(), u.clone(), fc.clone()   # consumers on the same stream
    side.synchronize()
    assert_planes_match((p2.cpu().numpy(), u2.cpu().numpy(), fc2.cpu().numpy()), want, ref_row=g.encoded_ref_group,
                        what="side stream")
    got = engine.run_dense(base, 0, 256, device_out=True)  # back on the default stream
    torch.cuda.synchronize()
    assert_planes_match(tuple(t.cpu().numpy() for t in got), want, ref_row=g.encoded_ref_group, what="default stream")


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_deferred_dense_calls_complete_their_leftover_genes(engine, test):
    """ILLICO_FLAG_DEFER: device input + device planes, no host round trip inside the call.  Genes the fused single-pass route
    cannot take (values of 64 and more, fractional values) are recomputed when the NEXT call or synchronize() looks at the
    route flags: alternating planes (the next pass is enqueued first), the same planes twice (completed before), a change of
    groups in between, and a sparse call in between -- every variant ends with the planes of the plain call, byte for byte."""
    import torch
    X, rng = make_counts(41, 6000, 320, 0.5)
    X[:, 7] = rng.poisson(70.0, size=6000)                       # beyond the 64-value table
    X[:, 100] = np.log1p(X[:, 100] * rng.uniform(0.5, 1.5, 6000))  # continuous
    X[5, 200] = 0.5
    X2 = X[:, ::-1].copy()
    labels = make_labels(rng, 6000, 30, n_ref=500)
    ref = "non-targeting" if test == "ovo" else None
    _, g = oracle.encode_and_count_groups(labels, ref)
    _, g2 = oracle.encode_and_count_groups(labels, "pert_00003" if test == "ovo" else None)
    Xd, X2d = torch.from_numpy(X).cuda(), torch.from_numpy(X2).cuda()
    engine.set_groups(g)
    want = [t.cpu().numpy() for t in engine.run_dense(Xd, 0, 320, device_out=True)]
    want2 = [t.cpu().numpy() for t in engine.run_dense(X2d, 0, 320, device_out=True)]
    assert_planes_match(want, oracle.run(X, g), ref_row=g.encoded_ref_group if ref else None, what="plain call")

    def planes():
        return tuple(torch.full((len(g.counts), 320), -7.0, dtype=torch.float64, device="cuda") for _ in range(3))

    def same(got, w):
        for a, b in zip(got, w):
            assert a.cpu().numpy().tobytes() == b.tobytes()

    # alternating planes: call 2 is enqueued before call 1 is completed
    A, B = planes(), planes()
    engine.run_dense(Xd, 0, 320, out=A, defer=True)
    engine.run_dense(X2d, 0, 320, out=B, defer=True)
    engine.run_dense(Xd, 0, 320, out=A, defer=True)
    engine.synchronize()
    same(A, want); same(B, want2)
    # the same planes again: the earlier call is completed first, then overwritten
    engine.run_dense(X2d, 0, 320, out=A, defer=True)
    engine.run_dense(Xd, 0, 320, out=A, defer=True)
    engine.synchronize()
    same(A, want)
    # a window, then new groups while a deferred call is pending (its leftovers need the old groups), then a sparse call
    C = planes()
    engine.run_dense(Xd, 64, 256, out=tuple(t[:, 64:256] for t in C), defer=True)
    engine.set_groups(g2)
    w3 = [t.cpu().numpy() for t in engine.run_dense(Xd, 0, 320, device_out=True)]
    engine.set_groups(g)
    for a, b in zip(C, want):
        assert a[:, 64:256].cpu().numpy().tobytes() == np.ascontiguousarray(b[:, 64:256]).tobytes()
        assert bool((a[:, :64] == -7.0).all()) and bool((a[:, 256:] == -7.0).all())
    assert_planes_match(w3, oracle.run(X, g2), ref_row=g2.encoded_ref_group if ref else None, what="other groups")
    D = planes()
    engine.run_dense(Xd, 0, 320, out=D, defer=True)
    M = sparse.csc_matrix(X[:, :40])
    sp = engine.run_sparse("csc", M.data, M.indices, M.indptr, M.shape, 0, 40)
    same(D, want)
    for a, b in zip(sp, want):
        np.testing.assert_array_equal(a, b[:, :40])
