"""Minimal AnnData stand-in: ``anndata`` is optional for this engine.

``asymptotic_wilcoxon`` only touches ``.X``, ``.layers``, ``.obs[key]`` and ``.var_names``
(reference illico/asymptotic_wilcoxon.py:178-206), which this class provides.
"""
from __future__ import annotations

import pandas as pd


class AnnDataLite:
    def __init__(self, X, obs: pd.DataFrame, var: pd.DataFrame | None = None, layers: dict | None = None):
        self.X = X
        self.obs = obs
        n_genes = X.shape[1]
        self.var = var if var is not None else pd.DataFrame(index=[f"gene_{i}" for i in range(n_genes)])
        self.layers = layers or {}
        self.isbacked = False

    @property
    def var_names(self):
        return self.var.index

    @property
    def shape(self):
        return self.X.shape

    def copy(self):
        X = self.X.copy() if hasattr(self.X, "copy") else self.X.clone()
        return AnnDataLite(X, self.obs.copy(), self.var.copy(), {k: v.copy() for k, v in self.layers.items()})
