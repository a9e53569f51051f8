"""Drop-in for ``illico.asymptotic_wilcoxon`` (reference illico/asymptotic_wilcoxon.py:71-258).

Same signature, same DataFrame (MultiIndex (pert, feature); float64 columns p_value, statistic,
fold_change; group-major rows).  The body is a ctypes driver of the MI355X engine: the gene-chunk
loop of the reference (asymptotic_wilcoxon.py:213-249) becomes calls of the registered dispatcher,
which writes its [G, w] planes straight into the result arrays.
"""
from __future__ import annotations

import math
import threading
from typing import Literal

import numpy as np
import pandas as pd
from scipy import sparse

from illico_amd.utils.groups import GroupContainer, encode_and_count_groups
from illico_amd.utils.registry import DataHandler, Test, data_handler_registry, dispatcher_registry

__all__ = ["asymptotic_wilcoxon", "operator"]

#: host bytes of one streamed gene chunk of a backed input (illico_amd/streaming.py: the chunk being read plus two pinned
#: staging slots are alive at any time)
STREAM_CHUNK_BYTES = 256 << 20


def _csr_to_device(X, data_handler, check=True):
    """(indices sorted?, handler of a device-resident copy) of an in-RAM scipy CSR matrix."""
    import torch
    from illico_amd._lib import get_engine
    from illico_amd.utils.registry import CSRDataHandler, CSRMatrix
    eng = get_engine()
    dev = torch.device("cuda", eng.device)
    from illico_amd._lib import normalize_values
    # the same dtype rules as the host path of Engine.run_sparse: values widened losslessly, one index dtype for both arrays
    idt = np.int32 if (X.indices.dtype == np.int32 and X.indptr.dtype == np.int32) else np.int64
    host = (normalize_values(np.asarray(X.data)), np.asarray(X.indices, dtype=idt), np.asarray(X.indptr, dtype=idt))
    d, i, p = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in host)
    sorted_ok = eng.csr_indices_sorted(i, p, X.shape[0]) if check else True
    return sorted_ok, CSRDataHandler(CSRMatrix(d, i, p, X.shape))


def operator(data_handler: DataHandler, lb: int, ub: int, group_container: GroupContainer, is_log1p: bool,
             use_continuity: bool, alternative: str, tie_correct: bool, out=None):
    """One gene chunk -- the reference's ``operator`` (asymptotic_wilcoxon.py:29-68), not delayed."""
    test = Test.OVR if group_container.encoded_ref_group == -1 else Test.OVO
    dispatcher = dispatcher_registry.get(test, data_handler.kernel_data_format())
    if lb < 0 or ub > data_handler.data.shape[1] or lb > ub:
        raise ValueError(f"Invalid chunk bounds: {(lb, ub)} for data with {data_handler.data.shape[1]} columns.")
    fetched_data, bounds = data_handler.fetch(lb, ub)
    X = data_handler.to_nb(fetched_data)
    kw = {} if out is None else {"out": out}
    pvalues, statistics, fold_change = dispatcher(X, *bounds, group_container, is_log1p, use_continuity, tie_correct,
                                                  alternative, **kw)
    return (pvalues, statistics, fold_change), (lb, ub)


def asymptotic_wilcoxon(
    adata,
    is_log1p: bool,
    group_keys: str,
    reference: str | None = None,
    n_threads: int = 1,
    batch_size: int | Literal["auto"] = "auto",
    alternative: str = "two-sided",
    use_continuity: bool = True,
    tie_correct: bool = True,
    layer: str | None = None,
    precompile: bool = True,
) -> pd.DataFrame:
    """Asymptotic Mann-Whitney / Wilcoxon rank-sum tests per (group, gene) on one MI355X.

    Parameters are those of the reference (asymptotic_wilcoxon.py:84-116).  ``adata`` is an
    ``anndata.AnnData`` or any object with ``.X``, ``.layers``, ``.obs[group_keys]`` and
    ``.var_names`` (``illico_amd.AnnDataLite``).  ``reference=None`` runs one-versus-rest.
    ``n_threads`` is accepted and ignored (one GPU context does the work); ``precompile`` is a
    no-op (HIP code objects are built ahead of time).  ``batch_size="auto"`` hands the whole gene
    range to the engine, which batches by HBM budget and computes every column (the reference's
    "auto" splitter leaves one column per chunk boundary uncomputed); an integer
    ``batch_size`` issues one dispatcher call per chunk like the reference.

    In an OVO call the reference-group row is (p=1.0, statistic=-1.0) in every format (the
    reference writes that in its sparse path, sparse_ovo.py:140-143, and leaves the row
    uninitialised in its dense path).

    Raises ``ValueError`` (unsorted CSR indices, unknown reference label, bad ``batch_size`` or
    ``alternative``) and ``KeyError`` (unsupported container) like the reference.
    """
    X = adata.layers[layer] if layer is not None else adata.X
    data_handler = data_handler_registry.get(X)

    if isinstance(X, sparse.csr_matrix) or (hasattr(sparse, "csr_array") and isinstance(X, sparse.csr_array)):
        # CSR rows span every gene, so the engine needs the whole matrix on the device anyway: it goes up once, here, and
        # both the sortedness check of the reference (asymptotic_wilcoxon.py:186-193) and every chunk run on that copy
        # (checking 2e8 indices on the host costs as much as the upload + the whole device computation).
        # (Small matrices are checked on the host first: argument errors then surface before any device work.)
        if X.nnz <= 1_000_000:
            from illico_amd.utils.ranking import check_indices_sorted_per_parcel
            sorted_ok = check_indices_sorted_per_parcel(X.indices, X.indptr)
            if sorted_ok:
                _, data_handler = _csr_to_device(X, data_handler, check=False)
        else:
            sorted_ok, data_handler = _csr_to_device(X, data_handler)
        if not sorted_ok:
            raise ValueError(
                "Input data matrix indices are not sorted. This is very unusual and may lead to incorrect results. "
                "This can be the result of operations like `adata[:, np.random.choice(…)]` that do not preserve sorting."
                "Please make sure that indices used to chunk the adata or the expression matrix have been sorted "
                "prior to computing DE genes.")
    if alternative not in ("two-sided", "less", "greater"):
        raise ValueError(f"Unsupported alternative hypothesis: {alternative}")

    raw_groups = adata.obs[group_keys]  # a categorical column is encoded from its codes (no pass over N strings)
    unique_raw_groups, group_container = encode_and_count_groups(groups=raw_groups, ref_group=reference)
    n_genes = X.shape[1]
    n_groups = int(group_container.counts.size)

    streams = bool(getattr(data_handler, "streams", False))
    if streams and (batch_size == "auto" or n_genes < 256):
        # out-of-core input: stream gene chunks of about STREAM_CHUNK_BYTES from storage (registry.py:162-188)
        per_gene = max(1, X.shape[0] * getattr(getattr(X, "dtype", None), "itemsize", 4))
        w = int(max(1, min(n_genes, STREAM_CHUNK_BYTES // per_gene)))
        bounds = np.append(np.arange(0, n_genes, w), n_genes)
        iterator = list(zip(bounds[:-1].tolist(), bounds[1:].tolist()))
    elif n_genes < 256 or batch_size == "auto":
        iterator = [(0, n_genes)]
    elif isinstance(batch_size, (int, np.integer)) and not isinstance(batch_size, bool):
        bs = min(int(batch_size), math.ceil(n_genes / max(int(n_threads), 1)))
        if bs <= 0:
            raise ValueError(f"Invalid batch_size value: {batch_size}. Must be 'auto' or an integer.")
        bounds = np.append(np.arange(0, n_genes, bs), n_genes)
        iterator = list(zip(bounds[:-1].tolist(), bounds[1:].tolist()))
    else:
        raise ValueError(f"Invalid batch_size value: {batch_size}. Must be 'auto' or an integer.")

    # three [G, n_genes] planes; each chunk writes its [:, lb:ub] window in place
    planes = np.empty((3, n_groups, n_genes), dtype=np.float64)
    iterator = [(lb, ub) for lb, ub in iterator if ub > lb]
    # the G x M row index of the result (asymptotic_wilcoxon.py:252-256) does not depend on the statistics: it is built on a thread of
    # its own while the engine works (numpy releases the GIL in repeat / tile, ctypes releases it in the engine calls) -- 15 - 30 ms of
    # the call at 2000 x 8000 that used to follow the pass
    cols = pd.Series(np.asarray(adata.var_names), name="feature", dtype=str)
    rows = pd.Series(unique_raw_groups, name="pert", dtype=str)
    index_box: list = []

    def build_index():
        try:
            index_box.append(_product_index(rows, cols))
        except BaseException as e:  # re-raised on the caller's thread
            index_box.append(e)

    index_thread = threading.Thread(target=build_index, name="illico-index", daemon=True)
    index_thread.start()
    if streams and len(iterator) > 1:
        from illico_amd.streaming import run_streaming
        run_streaming(data_handler, iterator, group_container, is_log1p, use_continuity, alternative, tie_correct, planes)
    else:
        for lb, ub in iterator:
            out = tuple(planes[k][:, lb:ub] for k in range(3))
            operator(data_handler, lb, ub, group_container, is_log1p, use_continuity, alternative, tie_correct, out=out)

    index_thread.join()
    if isinstance(index_box[0], BaseException):
        raise index_box[0]
    return pd.DataFrame(
        {"p_value": planes[0].reshape(-1), "statistic": planes[1].reshape(-1), "fold_change": planes[2].reshape(-1)},
        index=index_box[0],
        copy=False,
    )


def _product_index(rows: pd.Series, cols: pd.Series) -> pd.MultiIndex:
    """``pd.MultiIndex.from_product([rows, cols], names=["pert", "feature"])`` (asymptotic_wilcoxon.py:252-256) -- the same sorted
    levels and the same codes --, with the G x M codes written once in the smallest integer type instead of being built as int64
    cartesian products and narrowed afterwards: at 2000 x 8000 the index is most of what the call costs besides the engine."""
    ri, ci = pd.Index(rows), pd.Index(cols)
    if not (ri.is_unique and ci.is_unique) or ri.hasnans or ci.hasnans or len(ri) == 0 or len(ci) == 0:
        return pd.MultiIndex.from_product([rows, cols], names=["pert", "feature"])
    lr, lc = ri.sort_values(), ci.sort_values()   # from_product's levels: the sorted distinct labels
    G, M = len(ri), len(ci)

    def small(n):
        return np.int8 if n < 128 else np.int16 if n < 32768 else np.int32 if n < 2**31 else np.int64

    rcode = np.repeat(lr.get_indexer(ri).astype(small(G)), M)
    ccode = np.tile(lc.get_indexer(ci).astype(small(M)), G)
    return pd.MultiIndex(levels=[lr, lc], codes=[rcode, ccode], names=["pert", "feature"], verify_integrity=False)
