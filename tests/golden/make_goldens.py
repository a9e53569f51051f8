"""Generate golden vectors by running the REFERENCE ITSELF (remydubois/illico @ /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  The reference is
pure Python + Numba; Numba/anndata/h5py/loguru are not installed here, so it is imported with its
own supported un-jitted mode (reference illico/utils/compile.py:28-30, NUMBA_DISABLE_JIT): four
tiny stand-in modules are put on sys.path at run time -- `numba` (njit = identity decorator),
`anndata` (attribute bag), `h5py` (placeholder type), `loguru` (silent logger).  No reference
source is copied: only inputs (seeded) and the reference's outputs are stored, as .npz fixtures.

Usage:  python tests/golden/make_goldens.py         (writes tests/golden/*.npz)
"""
from __future__ import annotations

import os
import sys
import tempfile
import textwrap
from pathlib import Path

import numpy as np
import pandas as pd
from scipy import sparse

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")

STUBS = {
    "numba/__init__.py": """
        def njit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            def deco(f):
                return f
            return deco
        jit = njit
        def set_num_threads(n):
            pass
        class _Any:
            def __getattr__(self, name):
                return _Any()
            def __getitem__(self, item):
                return _Any()
            def __call__(self, *a, **k):
                return _Any()
        types = _Any()
        """,
    "loguru/__init__.py": """
        class _L:
            def __getattr__(self, name):
                return lambda *a, **k: None
        logger = _L()
        """,
    "h5py/__init__.py": """
        class Dataset:
            pass
        """,
    "anndata/__init__.py": """
        from . import _core
        class AnnData:
            def __init__(self, X=None, obs=None, var=None, layers=None):
                self.X = X
                self.obs = obs
                self.var = var
                self.layers = layers or {}
                self.isbacked = False
            @property
            def var_names(self):
                return self.var.index
            @property
            def shape(self):
                return self.X.shape
        """,
    "anndata/_core/__init__.py": """
        from . import sparse_dataset
        """,
    "anndata/_core/sparse_dataset.py": """
        class _CSCDataset:
            pass
        class _CSRDataset:
            pass
        """,
}


def import_reference():
    tmp = tempfile.mkdtemp(prefix="illico_stubs_")
    for rel, src in STUBS.items():
        p = Path(tmp) / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(src))
    sys.path.insert(0, tmp)
    sys.path.insert(0, str(REF))
    os.environ["NUMBA_DISABLE_JIT"] = "1"
    sys.dont_write_bytecode = True
    import anndata  # noqa: F401  (stub)
    import illico  # noqa: F401  (the reference)
    from illico import asymptotic_wilcoxon
    return asymptotic_wilcoxon, anndata


def make_counts(seed, n_cells, n_genes, sparsity, dtype=np.float32):
    """Fixture recipe of the reference's tests/conftest.py:82-96 (seeded Poisson + random mask)."""
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


CASES = [
    # name, seed, cells, genes, groups, sparsity, n_ref(None=uniform groups), continuous
    ("c1_1k_200_10", 0, 1000, 200, 10, 0.5, 100, False),
    ("small_ragged", 1, 700, 33, 7, 0.5, None, False),
    ("sparse90", 2, 1500, 40, 12, 0.9, 60, False),
    ("continuous", 3, 600, 24, 6, 0.3, None, True),
]
SWEEP = [(alt, cc, tc) for alt in ("two-sided", "less", "greater") for cc in (True, False) for tc in (True, False)]


def main():
    asymptotic_wilcoxon, anndata = import_reference()
    from illico.utils.ranking import _accumulate_group_ranksums_from_argsort, rank_sum_and_ties_from_sorted
    from illico.utils.math import compute_pval

    for name, seed, n_cells, n_genes, n_groups, sparsity, n_ref, continuous in CASES:
        X, rng = make_counts(seed, n_cells, n_genes, sparsity)
        if continuous:
            X = np.log1p(X * rng.uniform(0.5, 1.5, size=X.shape)).astype(np.float32)
        labels = make_labels(rng, n_cells, n_groups, n_ref)
        reference = "non-targeting" if n_ref is not None else labels[0]
        out = {"X": X, "labels": labels, "reference": np.array(reference)}
        var = pd.DataFrame(index=[f"gene_{i}" for i in range(n_genes)])
        obs = pd.DataFrame({"pert": labels})
        for fmt in ("dense", "csc", "csr"):
            M = {"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}[fmt]
            for test in ("ovo", "ovr"):
                sweep = SWEEP if name in ("c1_1k_200_10", "small_ragged") and fmt == "dense" else SWEEP[:1]
                if name == "small_ragged":
                    sweep = SWEEP
                for alt, cc, tc in sweep:
                    adata = anndata.AnnData(X=M.copy(), obs=obs.copy(), var=var.copy())
                    df = asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert",
                                             reference=reference if test == "ovo" else None,
                                             n_threads=1, batch_size=16, alternative=alt, use_continuity=cc,
                                             tie_correct=tc, precompile=False)
                    G = df.index.get_level_values(0).nunique()
                    res = df.values.reshape(G, n_genes, 3)
                    key = f"{fmt}|{test}|{alt}|{int(cc)}|{int(tc)}"
                    out[key] = res.astype(np.float64)
                    if "groups" not in out:
                        out["groups"] = np.array(df.index.get_level_values(0).unique().tolist())
        # is_log1p=True fold change (only fold_change differs); dense f32 only
        adata = anndata.AnnData(X=np.log1p(X), obs=obs.copy(), var=var.copy())
        df = asymptotic_wilcoxon(adata, is_log1p=True, group_keys="pert", reference=reference, n_threads=1,
                                 batch_size=16, precompile=False)
        out["dense|ovo|log1p"] = df.values.reshape(-1, n_genes, 3).astype(np.float64)
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print("wrote", name, len(out), "arrays")

    # ---- primitive-level goldens (reference tests/utils/test_ranking.py:13-56 style inputs) ----
    prim = {}
    rng = np.random.RandomState(0)
    for t in range(8):
        nA, nB = rng.randint(0, 40), rng.randint(1, 30)
        A = np.sort(rng.randint(0, 10, size=nA)).astype(np.float64)
        B = np.sort(rng.randint(0, 10, size=nB)).astype(np.float64)
        rs, ts = rank_sum_and_ties_from_sorted(A, B)
        prim[f"merge{t}_A"], prim[f"merge{t}_B"], prim[f"merge{t}_out"] = A, B, np.array([rs, ts], dtype=np.float64)
    for t in range(4):
        n, G = 30 + 17 * t, 3 + t
        arr = rng.rand(n) if t % 2 == 0 else rng.randint(0, 6, size=n).astype(np.float64)
        groups = rng.randint(0, G, size=n)
        idx = np.argsort(arr, kind="stable")
        ranksums = np.zeros(G, dtype=np.float64)
        ts = _accumulate_group_ranksums_from_argsort(arr, idx, groups, ranksums)
        prim[f"acc{t}_arr"], prim[f"acc{t}_groups"], prim[f"acc{t}_idx"] = arr, groups, idx
        prim[f"acc{t}_ranksums"], prim[f"acc{t}_tie"] = ranksums, np.array([ts], dtype=np.float64)
    rows = []
    for alt in ("two-sided", "less", "greater"):
        for cc in (0.0, 0.5):
            for (n_ref, n_tgt, tie, U) in [(100, 90, 0.0, 4000.0), (100, 90, 5.0e5, 4500.0), (10000, 145, 3.0e11, 9.0e5),
                                           (10000, 145, 1.0e11, 2.0e5), (50, 3, 0.0, 75.0), (7, 7, 300.0, 0.0),
                                           (33333, 193, 2.1e13, 1.0e6), (5, 5, 990.0, 12.5)]:
                n = n_ref + n_tgt
                mu = n_ref * n_tgt / 2.0
                pv = compute_pval(n_ref, n_tgt, n, tie, U, mu, cc, alt)
                rows.append([n_ref, n_tgt, tie, U, cc, {"two-sided": 0, "less": 1, "greater": 2}[alt], pv])
    prim["pval_rows"] = np.array(rows, dtype=np.float64)
    np.savez_compressed(HERE / "primitives.npz", **prim)
    print("wrote primitives", len(prim))


def make_stored_zeros():
    """Explicitly stored zeros (and negatives) in sparse input: the reference's SPARSE kernels treat every stored entry as a
    value above the column's implicit zeros (illico/ovo/sparse_ovo.py:74, ovr/sparse_ovr.py:77), its DENSE kernels rank the
    same numbers as the numbers they are.  Both outputs of the reference are recorded for the same matrix -- the sparse one
    pins that behaviour, the dense one is what this repository returns for sparse input too."""
    asymptotic_wilcoxon, anndata = import_reference()
    rng = np.random.RandomState(11)
    n_cells, n_genes, n_groups = 400, 12, 5
    X = (rng.poisson(2.0, size=(n_cells, n_genes)) * (rng.rand(n_cells, n_genes) < 0.4)).astype(np.float32)
    labels = make_labels(rng, n_cells, n_groups, 60)
    out = {"X": X, "labels": labels, "reference": np.array("non-targeting")}
    var = pd.DataFrame(index=[f"gene_{i}" for i in range(n_genes)])
    obs = pd.DataFrame({"pert": labels})
    M = sparse.csc_matrix(X)
    stored_zero = np.zeros(M.nnz, dtype=bool)
    stored_zero[::7] = True                      # every 7th stored entry becomes an explicit zero
    M.data[stored_zero] = 0.0
    Xz = M.toarray()                             # the same numbers, dense
    out["Xz"] = Xz
    out["csc_data"], out["csc_indices"], out["csc_indptr"] = M.data.copy(), M.indices.copy(), M.indptr.copy()
    for test in ("ovo", "ovr"):
        for fmt, A in (("dense", Xz), ("csc", M.copy()), ("csr", sparse.csr_matrix((M.data, M.indices, M.indptr), shape=M.shape[::-1]).T.tocsr()
                                                          if False else None)):
            if A is None:
                continue
            adata = anndata.AnnData(X=A, obs=obs.copy(), var=var.copy())
            df = asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", reference="non-targeting" if test == "ovo" else None,
                                     n_threads=1, batch_size=16, precompile=False)
            G = df.index.get_level_values(0).nunique()
            out[f"{fmt}|{test}"] = df.values.reshape(G, n_genes, 3).astype(np.float64)
    np.savez_compressed(HERE / "stored_zeros.npz", **out)
    print("wrote stored_zeros", len(out), "arrays")


if __name__ == "__main__":
    if "--stored-zeros" in sys.argv:
        make_stored_zeros()
    else:
        main()
        make_stored_zeros()
