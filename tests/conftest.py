import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def make_counts(seed, n_cells, n_genes, sparsity, dtype=np.float32):
    """Seeded Poisson + mask recipe (same recipe as tests/golden/make_goldens.py::make_counts)."""
    rng = np.random.RandomState(seed)
    gene_means = rng.uniform(0.1, 15, size=n_genes)
    X = rng.poisson(gene_means, size=(n_cells, n_genes)).astype(dtype)
    mask = rng.rand(n_cells, n_genes) < sparsity
    X[mask] = 0
    return X, rng


def make_labels(rng, n_cells, n_groups, n_ref=None):
    if n_ref is None:
        codes = rng.randint(0, n_groups, size=n_cells)
        return np.array([f"pert_{g}" for g in codes])
    codes = np.concatenate([np.zeros(n_ref, dtype=int), 1 + rng.randint(0, n_groups - 1, size=n_cells - n_ref)])
    rng.shuffle(codes)
    return np.array(["non-targeting" if c == 0 else f"pert_{c:05d}" for c in codes])


def load_golden(name):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)


def assert_planes_match(got, want, *, ref_row=None, p_rtol=1e-12, fc_rtol=1e-12, what=""):
    """The reference's tolerance triple (reference tests/test_asymptotic_wilcoxon.py:166-185):
    statistic exact, p-value rtol 1e-12 / atol 0; fold change tightened from 1e-6 to 1e-12."""
    gp, gu, gfc = got
    wp, wu, wfc = want
    mask = np.ones(gp.shape[0], dtype=bool)
    if ref_row is not None and ref_row >= 0:
        mask[ref_row] = False  # reference leaves this row unspecified
    np.testing.assert_array_equal(gu[mask], wu[mask], err_msg=f"statistic {what}")
    np.testing.assert_allclose(gp[mask], wp[mask], rtol=p_rtol, atol=0.0, err_msg=f"p_value {what}")
    np.testing.assert_allclose(gfc, wfc, rtol=fc_rtol, atol=0.0, equal_nan=True, err_msg=f"fold_change {what}")
