"""Host-side checks that belong to the path (reference illico/utils/ranking.py:223-273)."""
from __future__ import annotations

import numpy as np


def check_indices_sorted_per_parcel(indices, indptr) -> bool:
    """True iff the column indices of every CSR row are non-decreasing (ranking.py:245-273).

    Vectorised: a descent ``indices[i] < indices[i-1]`` only counts when i-1 and i lie in the same row.
    """
    indices = np.asarray(indices)
    indptr = np.asarray(indptr)
    if indices.size < 2:
        return True
    desc = np.flatnonzero(indices[1:] < indices[:-1]) + 1  # positions i with a descent from i-1
    if desc.size == 0:
        return True
    starts = indptr[1:-1] if indptr.size > 2 else indptr[:0]
    # a descent at i is harmless iff i is the first entry of a row
    return bool(np.isin(desc, starts).all())
