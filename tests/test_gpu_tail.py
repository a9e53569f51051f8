"""The far tail of sparse OVR: p ~ 1e-290 (z ~ 36.5), where a relative error of the tie sum is amplified by ~z^2.

The reference's sparse OVR path forms the tie sum in float64 -- non-zero tie blocks first, then `n0**3 - n0` with n0 a float64
(illico/ovr/sparse_ovr.py:49,83) -- while its dense path adds exact integers (utils/ranking.py:31-47).  With two million cells and
97 % zeros n0^3 exceeds 2^53, so how the zero block is formed matters in the last bits of the tie sum, and those bits carry into p
at z ~ 36: this test holds every sparse OVR route to the oracle (the reference's arithmetic) at rtol 1e-12 exactly there."""
import numpy as np
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match

pytestmark = pytest.mark.gpu

N = 2_000_000


def _labels(rng, n_small, n_large=2):
    """`n_small` cells that will sit on top of everything, and `n_large` large groups."""
    codes = np.concatenate([np.zeros(n_small, dtype=int), 1 + rng.randint(0, n_large, size=N - n_small)])
    rng.shuffle(codes)
    return np.array([f"g{c}" for c in codes]), codes


def _matrix(rng, codes, continuous, n_genes=3):
    """97 % zeros; thousands of tie blocks among the stored values; the small group's cells hold the largest values of every gene."""
    X = np.zeros((N, n_genes), dtype=np.float32)
    for j in range(n_genes):
        nz = rng.rand(N) < 0.03
        if continuous:
            v = np.round(np.exp(rng.randn(N) * 0.8) + 0.05, 3).astype(np.float32)   # ~5000 distinct values: thousands of tie blocks
        else:
            v = (1 + rng.poisson(3.0 + j, size=N)).astype(np.float32)
        X[:, j] = np.where(nz, v, 0)
        top = codes == 0
        X[top, j] = (40.0 + (rng.rand(int(top.sum())) * 9).round(2 if continuous else 0)).astype(np.float32)
    return X


@pytest.fixture(scope="module")
def engine():
    from illico_amd._lib import get_engine
    return get_engine()


@pytest.mark.parametrize("fmt", ["csc", "csr"])
@pytest.mark.parametrize("continuous", [True, False])
def test_sparse_ovr_far_tail_matches_the_reference_arithmetic(engine, fmt, continuous):
    rng = np.random.RandomState(1234 + int(continuous))
    # n_small chosen so that z of the small group lands at ~36.5: z ~ sqrt(3 n_small / tie_corr), tie_corr ~ 1 - 0.97^3
    labels, codes = _labels(rng, 39)
    X = _matrix(rng, codes, continuous)
    _, g = oracle.encode_and_count_groups(labels, None)
    M = (sparse.csc_matrix if fmt == "csc" else sparse.csr_matrix)(X)
    want = oracle.run(M, g)
    p_small = want[0][0]
    assert np.all((p_small > 1e-305) & (p_small < 1e-250)), p_small   # the regime this test is about
    engine.set_groups(g)
    got = engine.run_sparse(fmt, M.data, M.indices, M.indptr, M.shape, 0, M.shape[1])
    assert_planes_match(got, want, what=f"far tail {fmt} continuous={continuous}")
    if fmt == "csr" and not continuous:   # the other count routes a CSR matrix can take: byte windows + the fused dense kernels; the transposition
        for opts in (dict(no_csr_counts_path=1), dict(no_csr_counts_path=1, no_dense_window_path=1)):
            for k, v in opts.items():
                engine.set_option(k, v)
            try:
                got = engine.run_sparse(fmt, M.data, M.indices, M.indptr, M.shape, 0, M.shape[1])
            finally:
                for k in opts:
                    engine.set_option(k, 0)
            assert_planes_match(got, want, what=f"far tail csr {opts}")


def test_dense_count_ovr_far_tail_matches_the_reference_arithmetic(engine):
    """The dense path of the reference adds the tie blocks' exact integers one after the other in float64, zeros first: beyond 2^53 every
    later block is rounded into the running sum.  The fused dense OVR kernels form the same float64 (FusedParams::tie_mode = 1)."""
    import torch
    rng = np.random.RandomState(77)
    labels, codes = _labels(rng, 39, n_large=40)   # (groups of ~50 000 cells: the fused route holds groups of up to 65 535)
    X = _matrix(rng, codes, False)
    _, g = oracle.encode_and_count_groups(labels, None)
    want = oracle.run(X, g)
    assert np.all((want[0][0] > 1e-305) & (want[0][0] < 1e-250)), want[0][0]
    engine.set_groups(g)
    engine.set_option("profile", 1)
    engine.profile_reset()
    try:
        got = engine.run_dense(torch.from_numpy(X).cuda(), 0, X.shape[1])   # device-resident: the fused single-pass route
        prof = engine.profile_get()
    finally:
        engine.set_option("profile", 0)
    assert ("k_ovr_fused" in prof or "k_group_value_hists" in prof) and "k_ovr_counts" not in prof, prof   # (39 groups of ~50 000 cells: the group-histogram form of the fused route)
    assert_planes_match(got, want, what="far tail dense counts")
