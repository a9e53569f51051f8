"""Values at the ends of their type on every dense route: the largest / smallest int32 and int64 (the largest one's sortable key
is the all-ones pattern the kernels also use as "empty slot" and padding), +-infinity in float32 / float64 -- in a few cells
of a column (a handful of equal keys in the reference's last bucket) and in a third of a column (tie-heavy: the routes hand
such genes on).  Statistic exact, p-value and fold change at rtol 1e-12 against the oracle.  The reference ranks them like any
other value (np.sort based, illico/utils/ranking.py:52-158)."""
import numpy as np
import pytest

import oracle
from conftest import assert_planes_match, make_labels

pytestmark = pytest.mark.gpu

ROUTES = [{}, {"no_fused_path": 1}, {"no_fused_path": 1, "no_packed_dense": 1},
          {"no_fused_path": 1, "no_packed_dense": 1, "no_ovo_ref_buckets": 1, "no_ovr_parts_path": 1}]


@pytest.mark.parametrize("dtype", [np.int32, np.int64, np.float32, np.float64])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_type_extremes_on_every_dense_route(dtype, test):
    from illico_amd._lib import get_engine
    eng = get_engine()
    rng = np.random.RandomState(3)
    n, m = 4000, 8
    labels = make_labels(rng, n, 30, n_ref=800)
    integer = np.issubdtype(dtype, np.integer)
    big = np.iinfo(dtype).max if integer else np.inf
    small = np.iinfo(dtype).min if integer else -np.inf
    X = (rng.poisson(3.0, size=(n, m)) * rng.randint(1, 1000, size=(n, m))).astype(dtype)
    X[rng.rand(n, m) < 0.5] = 0
    X[rng.rand(n) < 0.01, 0] = big
    X[rng.rand(n) < 0.3, 1] = big
    X[rng.rand(n) < 0.05, 2] = small
    if dtype != np.int64:  # (+-2^63 in one column: the value sums cancel to rounding noise, in the reference's float64 as well)
        X[rng.rand(n) < 0.01, 3] = big
        X[rng.rand(n) < 0.01, 3] = small
    _, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    want = oracle.run(np.ascontiguousarray(X, dtype=np.float64), g)
    for opt in ROUTES:
        for k, v in opt.items():
            eng.set_option(k, v)
        try:
            eng.set_groups(g)
            got = eng.run_dense(X, 0, m)
        finally:
            for k in opt:
                eng.set_option(k, 0)
        # fold changes of the columns that hold an infinity (or sums beyond float64's integers) are inf / nan on both sides
        assert_planes_match(got, want, ref_row=g.encoded_ref_group if test == "ovo" else None, what=f"{np.dtype(dtype).name} {test} {opt}")
