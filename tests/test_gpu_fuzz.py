"""Randomised shapes / formats / dtypes / options through every input path of the engine vs the CPU oracle."""
import os

import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match

pytestmark = pytest.mark.gpu


def _env_seeds():
    """ILLICO_FUZZ_SEEDS="40-99,106-130,206-260": more seeds for a one-off sweep (below 100: small cases, from 100: large ones, from 200: heavy-tailed counts among them)."""
    out = []
    for part in filter(None, os.environ.get("ILLICO_FUZZ_SEEDS", "").split(",")):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


@pytest.fixture(scope="module")
def engine():
    """ILLICO_FUZZ_OPTIONS="no_fused_path=1,packed_eq_buckets=1": engine options for the whole sweep (forces routes)."""
    from illico_amd._lib import get_engine
    eng = get_engine()
    opts = [kv.split("=") for kv in filter(None, os.environ.get("ILLICO_FUZZ_OPTIONS", "").split(","))]
    for k, v in opts:
        eng.set_option(k, int(v))
    yield eng
    for k, _ in opts:
        eng.set_option(k, -1 if k == "packed_eq_buckets" else 0)


def _case(seed):
    rng = np.random.RandomState(1000 + seed)
    if seed >= 100:   # a few larger cases: several gene tiles and batches, hundreds of groups, groups above 255 cells
        n, m, G = 20000, int(rng.choice([520, 700])), int(rng.choice([30, 300]))
    else:
        n = int(rng.choice([37, 150, 600, 2500]))
        m = int(rng.choice([1, 7, 64, 65, 130, 300]))
        G = int(rng.randint(2, min(40, n // 2) + 1))
    # (from seed 200: also heavy-tailed counts -- log-normal gene means: most 64-gene tiles hold a gene beyond the 64-value table, a few
    # genes go beyond 255 / 2047 / 4095: the 256-value stage on gathered columns, the histogram kernels, the sort routes)
    kind = rng.choice(["counts", "counts-large", "continuous", "mixed"] + (["counts-heavy"] * 3 if seed >= 200 else []))
    density = float(rng.choice([0.02, 0.1, 0.5, 1.0]))
    mask = rng.rand(n, m) < density
    if kind == "counts":
        X = rng.poisson(rng.uniform(0.2, 8.0, size=m), size=(n, m)) * mask
    elif kind == "counts-large":
        X = rng.poisson(rng.uniform(0.2, 90.0, size=m), size=(n, m)) * mask
    elif kind == "counts-heavy":
        X = rng.poisson(np.exp(rng.normal(2.0, 1.9, size=m)).clip(0.05, 5000.0), size=(n, m)) * mask
    elif kind == "continuous":
        X = np.log1p(rng.poisson(3.0, size=(n, m)) * rng.uniform(0.5, 1.5, size=(n, m))) * mask
    else:
        X = (rng.poisson(2.0, size=(n, m)) * mask).astype(np.float64)
        bad = rng.rand(m) < 0.3
        X[:, bad] = X[:, bad] * rng.uniform(0.5, 1.5, size=(n, int(bad.sum())))
    sizes = rng.multinomial(n - G, rng.dirichlet(np.ones(G) * 0.7)) + 1
    labels = np.concatenate([[f"g{i:03d}"] * int(s) for i, s in enumerate(sizes)])
    rng.shuffle(labels)
    dtype = rng.choice([np.float32, np.float64, np.int32, np.int64]) if kind.startswith("counts") else rng.choice([np.float32, np.float64])
    X = X.astype(dtype).astype(np.float64)
    lb = int(rng.randint(0, m))
    ub = int(rng.randint(lb + 1, m + 1))
    if kind == "counts-heavy":  # the whole width: enough tiles for the device to leave the 256-value stage to the gathered columns
        lb, ub = 0, m
    opts = dict(use_continuity=bool(rng.randint(2)), tie_correct=bool(rng.randint(2)),
                alternative=str(rng.choice(["two-sided", "less", "greater"])),
                is_log1p=bool(kind == "continuous" and rng.rand() < 0.5))
    ref = str(rng.choice(np.unique(labels))) if rng.rand() < 0.6 else None
    return X, dtype, labels, ref, lb, ub, opts, kind


@pytest.mark.parametrize("seed", list(range(40)) + [100, 101, 102, 103, 104, 105, 200, 201, 202, 203, 204, 205] + _env_seeds())
def test_random_case_all_input_paths(engine, seed):
    import torch
    X, dtype, labels, ref, lb, ub, opts, kind = _case(seed)
    _, g = oracle.encode_and_count_groups(labels, ref)
    want = oracle.run(X, g, col_lb=lb, col_ub=ub, **opts)
    engine.set_groups(g)
    Xt = np.ascontiguousarray(X.astype(dtype))
    rr = g.encoded_ref_group if ref is not None else None
    # expm1 is taken in the input dtype (math.py:212): for float32 input the device's expm1f and libm's may differ in the
    # last float32 bit, so that case is held to the reference's own tolerance; everything else to 1e-12
    fc = 1e-6 if (opts["is_log1p"] and np.dtype(dtype) == np.float32) else 1e-12
    what = f"seed {seed} {kind} {np.dtype(dtype).name} {X.shape} ref={ref} [{lb},{ub}) {opts}"
    got = engine.run_dense(Xt, lb, ub, **opts)
    assert_planes_match(got, want, ref_row=rr, fc_rtol=fc, what="dense-host " + what)
    got = engine.run_dense(torch.from_numpy(Xt).cuda(), lb, ub, **opts)
    assert_planes_match(got, want, ref_row=rr, fc_rtol=fc, what="dense-device " + what)
    for fmt, ctor in (("csc", sparse.csc_matrix), ("csr", sparse.csr_matrix)):
        M = ctor(Xt)
        got = engine.run_sparse(fmt, M.data, M.indices, M.indptr, M.shape, lb, ub, **opts)
        assert_planes_match(got, want, ref_row=rr, fc_rtol=fc, what=f"{fmt}-host " + what)
        d, i, p = (torch.from_numpy(a).cuda() for a in (M.data, M.indices, M.indptr))
        got = engine.run_sparse(fmt, d, i, p, M.shape, lb, ub, **opts)
        assert_planes_match(got, want, ref_row=rr, fc_rtol=fc, what=f"{fmt}-device " + what)
