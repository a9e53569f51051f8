"""The reference's own end-to-end test, run against this engine: same fixture recipe (tests/conftest.py:76-100 of the
reference: 10 000 cells x 15 genes x 5 groups, Poisson(gene mean ~ U(0.1, 15)) float32, 50 % zeroed, RandomState(0)),
same parameter sweep and call (tests/test_asymptotic_wilcoxon.py:111-147: batch_size=16, n_threads=1), same judge
(scipy.stats.mannwhitneyu, asymptotic) and the same tolerances (:166-185: statistic exact, p-value rtol 1e-12,
fold change rtol 1e-6), plus its "input is not modified" check (:187-194)."""
import numpy as np
import pandas as pd
import pytest
from scipy import sparse
from scipy.stats import mannwhitneyu

pytestmark = pytest.mark.gpu


def _rand_adata(fmt):
    from illico_amd import AnnDataLite
    n_cells, n_genes, n_groups = 10_000, 15, 5
    rng = np.random.RandomState(0)
    gene_means = rng.uniform(0.1, 15, size=n_genes)
    dense = rng.poisson(gene_means, size=(n_cells, n_genes)).astype(np.float32)
    dense[rng.rand(n_cells, n_genes) < 0.5] = 0
    groups = rng.randint(0, n_groups, size=n_cells)
    X = {"dense": dense, "csc": sparse.csc_matrix(dense), "csr": sparse.csr_matrix(dense)}[fmt]
    obs = pd.DataFrame({"pert": [f"group_{g}" for g in groups]})
    return AnnDataLite(X, obs=obs), dense, obs["pert"].values


@pytest.mark.parametrize("fmt", ["dense", "csc", "csr"])
@pytest.mark.parametrize("alternative", ["two-sided", "less", "greater"])
@pytest.mark.parametrize("tie_correct", [True, False], ids=["tie-correct", "no-tie-correct"])
@pytest.mark.parametrize("use_continuity", [True, False])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_asymptotic_wilcoxon_like_the_reference(fmt, test, use_continuity, tie_correct, alternative):
    from illico_amd import asymptotic_wilcoxon
    adata, dense, labels = _rand_adata(fmt)
    cached = dense.copy()
    reference = labels[0] if test == "ovo" else None
    res = asymptotic_wilcoxon(adata=adata, is_log1p=False, group_keys="pert", reference=reference, use_continuity=use_continuity,
                              tie_correct=tie_correct, n_threads=1, batch_size=16, alternative=alternative)
    assert list(res.columns) == ["p_value", "statistic", "fold_change"] and len(res) == 5 * 15
    if tie_correct:  # SciPy always tie-corrects (the reference skips the comparison otherwise, :149-151)
        for lab in np.unique(labels):
            if lab == reference:
                continue
            grp = dense[labels == lab]
            rest = dense[labels == reference] if reference is not None else dense[labels != lab]
            st, pv = mannwhitneyu(rest, grp, axis=0, method="asymptotic", use_continuity=use_continuity, alternative=alternative)
            got = res.loc[lab]
            np.testing.assert_allclose(got.statistic.values, st, atol=0.0, rtol=0.0)
            np.testing.assert_allclose(got.p_value.values, pv, atol=0.0, rtol=1.0e-12)
            fc = grp.mean(axis=0, dtype=np.float64) / rest.mean(axis=0, dtype=np.float64)
            np.testing.assert_allclose(got.fold_change.values, fc, atol=0.0, rtol=1.0e-6)
    X = adata.X
    np.testing.assert_array_equal(X if isinstance(X, np.ndarray) else X.toarray(), cached)
