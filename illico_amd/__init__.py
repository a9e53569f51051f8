"""illico_amd -- MI355X-native engine for illico's asymptotic Wilcoxon rank-sum hot path.

``from illico_amd import asymptotic_wilcoxon`` is a drop-in for ``illico.asymptotic_wilcoxon``.
"""
from illico_amd.anndata_lite import AnnDataLite
from illico_amd.asymptotic_wilcoxon import asymptotic_wilcoxon

__all__ = ["asymptotic_wilcoxon", "AnnDataLite"]
__version__ = "0.1.0"
