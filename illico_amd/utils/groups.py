"""Group encoding -- same GroupContainer as the reference (illico/utils/groups.py:6-15, :18-58)."""
from __future__ import annotations

from collections import namedtuple
from typing import Any

import numpy as np

# same five fields, in the same order, as the reference's container (illico/utils/groups.py:6-15)
GroupContainer = namedtuple("GroupContainer", "encoded_groups counts indices indptr encoded_ref_group")


def encode_and_count_groups(groups, ref_group: Any):
    """Build the GroupContainer.

    Same outputs as the reference (labels ordered as ``np.unique`` orders them, int64 arrays,
    ``encoded_ref_group == -1`` for one-versus-rest), computed without the per-cell Python loop
    of groups.py:42-44: ``np.unique(return_inverse=True)`` gives the codes directly and a stable
    argsort of the codes gives ``indices`` sorted inside each group (the order the reference's
    own TODO at groups.py:46 asks for).
    """
    cat = getattr(groups, "cat", None)  # pandas categorical column (the usual dtype of AnnData.obs columns)
    if cat is not None and not groups.isna().any():
        return _encode_categorical(groups, ref_group)
    groups = np.asarray(groups)
    if groups.ndim == 1 and groups.dtype.kind in "USO" and groups.size >= 4096:
        fast = _encode_hashed(groups, ref_group)  # strings: one hash pass instead of a sort of N labels
        if fast is not None:
            return fast
    if ref_group is not None and not np.any(groups == ref_group):
        raise ValueError(f"Reference group `{ref_group}` is not present in the group labels.")
    unique_groups, encoded_groups, group_counts = np.unique(groups, return_inverse=True, return_counts=True)
    encoded_groups = np.ascontiguousarray(encoded_groups.reshape(-1), dtype=np.int64)
    group_counts = group_counts.astype(np.int64)
    group_indices = np.argsort(encoded_groups, kind="stable").astype(np.int64)
    group_indptr = np.concatenate([[0], np.cumsum(group_counts)]).astype(np.int64)
    if ref_group is None:
        encoded_ref = -1
    else:
        encoded_ref = int(np.flatnonzero(unique_groups == ref_group)[0])
    return unique_groups, GroupContainer(
        encoded_groups=encoded_groups,
        counts=group_counts,
        indices=group_indices,
        indptr=group_indptr,
        encoded_ref_group=encoded_ref,
    )


def _encode_categorical(col, ref_group: Any):
    """Same container as above from a pandas categorical column, without touching the N labels as strings:
    the categories that occur are ordered the way ``np.unique`` orders the labels, the per-cell codes are remapped."""
    cats = np.asarray(col.cat.categories)
    codes = np.asarray(col.cat.codes, dtype=np.int64)
    present = np.bincount(codes, minlength=cats.size) > 0
    order = np.argsort(cats, kind="stable")          # np.unique's order of the distinct labels
    order = order[present[order]]
    unique_groups = cats[order]
    if unique_groups.dtype == object:                # np.unique of a list of str gives a '<U' array
        unique_groups = np.array(unique_groups.tolist())
    if ref_group is not None and not np.any(unique_groups == ref_group):
        raise ValueError(f"Reference group `{ref_group}` is not present in the group labels.")
    remap = np.full(cats.size, -1, dtype=np.int64)
    remap[order] = np.arange(order.size)
    encoded_groups = remap[codes]
    group_counts = np.bincount(encoded_groups, minlength=order.size).astype(np.int64)
    group_indices = np.argsort(encoded_groups, kind="stable").astype(np.int64)
    group_indptr = np.concatenate([[0], np.cumsum(group_counts)]).astype(np.int64)
    encoded_ref = -1 if ref_group is None else int(np.flatnonzero(unique_groups == ref_group)[0])
    return unique_groups, GroupContainer(encoded_groups, group_counts, group_indices, group_indptr, encoded_ref)


def _encode_hashed(groups: np.ndarray, ref_group: Any):
    """The container of ``encode_and_count_groups`` for a 1-D array of string labels: ``pandas.factorize`` (a hash table: one pass
    over the N labels) finds the distinct labels, which are then ordered as ``np.unique`` orders them (a sort of G labels instead of
    N).  Returns None -- the caller takes the ``np.unique`` path -- for anything that is not plainly strings (missing values, mixed
    objects), so that errors and orderings of odd inputs stay exactly numpy's."""
    try:
        import pandas as pd
    except ImportError:  # pragma: no cover
        return None
    if groups.dtype.kind == "O" and not all(isinstance(x, str) for x in groups[:: max(1, groups.size // 1024)]):
        return None
    codes, uniques = pd.factorize(groups, sort=False)
    if codes.min(initial=0) < 0:
        return None
    uniques = np.asarray(uniques)
    if groups.dtype.kind == "O":
        if not all(isinstance(x, str) for x in uniques):
            return None
        uniques = np.array(uniques.tolist())  # np.unique of a list of str gives a '<U' array; same order either way
    else:
        uniques = uniques.astype(groups.dtype, copy=False)
    order = np.argsort(uniques, kind="stable")
    unique_groups = uniques[order]
    if ref_group is not None and not np.any(unique_groups == ref_group):
        raise ValueError(f"Reference group `{ref_group}` is not present in the group labels.")
    remap = np.empty(order.size, dtype=np.int64)
    remap[order] = np.arange(order.size)
    encoded_groups = remap[codes]
    group_counts = np.bincount(encoded_groups, minlength=order.size).astype(np.int64)
    group_indices = np.argsort(encoded_groups, kind="stable").astype(np.int64)
    group_indptr = np.concatenate([[0], np.cumsum(group_counts)]).astype(np.int64)
    encoded_ref = -1 if ref_group is None else int(np.flatnonzero(unique_groups == ref_group)[0])
    return unique_groups, GroupContainer(encoded_groups, group_counts, group_indices, group_indptr, encoded_ref)
