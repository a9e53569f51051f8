"""Gene sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Genes are independent units (every reference kernel takes an arbitrary contiguous gene range,
illico/asymptotic_wilcoxon.py:213-241), so rank r of R computes a contiguous gene range with no input
exchange; the only collective is the gather of the three float64 result planes to rank 0, issued per
gene block so that it overlaps the next block's compute.

Two tails for the drop-in call (whose consumer is a host DataFrame): the default brings every rank's planes to the host over that
rank's OWN PCIe link, straight into its column range of ONE [3][G][M] result in POSIX shared memory (`SharedHostPlanes`) -- the
reference's workers write their slices of one host result the same way, illico/asymptotic_wilcoxon.py:236-244; 8 links x ~55 GB/s
instead of 7 peers' planes over xGMI into rank 0 and then 3.6 GB (configs[4]) through rank 0's one link.  The device tail (RCCL p2p
gather into one device tensor on rank 0) stays for consumers that want the planes on a GPU.  `asymptotic_wilcoxon_threads` is the
single-process form (one host thread + one context per device, no process group at all: SURVEY.md 8e allows either).

Ranges are balanced by gene COUNT for dense input and by STORED ENTRIES for sparse input (SURVEY.md 8e:
"for sparse inputs balance by nnz rather than gene count"; the unit of work of the reference's sparse
kernels is the stored entry, illico/ovo/sparse_ovo.py:163-210).
"""
from __future__ import annotations

import mmap
import os
import threading
import uuid

import numpy as np


def shard_bounds(n: int, parts: int) -> list[tuple[int, int]]:
    """Contiguous, balanced (+-1) split of range(n) into EXACTLY `parts` windows (empty ones when n < parts: every
    rank must issue the same number of gathers whatever its own width)."""
    parts = max(1, int(parts))
    edges = [(n * i) // parts for i in range(parts + 1)]
    return [(edges[i], edges[i + 1]) for i in range(parts)]


def rank_gene_range(n_genes: int, rank: int, world: int) -> tuple[int, int]:
    return ((n_genes * rank) // world, (n_genes * (rank + 1)) // world)


def balanced_gene_ranges(weights, world: int) -> list[tuple[int, int]]:
    """Contiguous gene ranges, one per rank, whose weight sums are as equal as contiguity allows.

    ``weights[j]`` is gene j's cost (stored entries, plus a constant per gene if the caller wants one).  Rank r's range
    ends at the first gene where the running weight reaches r+1 shares of the total -- every rank computes the same
    edges from the same array, no communication.  Zero total weight falls back to the split by gene count.
    """
    w = np.asarray(weights, dtype=np.float64)
    m = int(w.size)
    world = max(1, int(world))
    total = float(w.sum())
    if m == 0 or total <= 0.0:
        return [rank_gene_range(m, r, world) for r in range(world)]
    cum = np.cumsum(w)
    # edge r = number of genes whose running weight stays at or below r shares (the gene that crosses a share goes to
    # whichever side leaves the smaller error)
    targets = total * np.arange(1, world, dtype=np.float64) / world
    hi = np.searchsorted(cum, targets, side="left")  # first gene with cum >= target
    edges = [0]
    for t, j in zip(targets, hi.tolist()):
        j = min(j, m - 1)
        below = cum[j - 1] if j > 0 else 0.0
        e = j + 1 if (cum[j] - t) <= (t - below) else j
        edges.append(int(min(max(e, edges[-1]), m)))
    edges.append(m)
    return [(edges[r], edges[r + 1]) for r in range(world)]


def sparse_gene_weights(X, per_gene_cost: float = 0.0) -> np.ndarray:
    """Stored entries per gene (column) of a scipy CSC / CSR matrix (or anything with indptr / indices / shape / format),
    plus ``per_gene_cost``.  CSC: differences of ``indptr`` (no pass over the entries); CSR: one ``bincount`` of the
    column indices."""
    fmt = getattr(X, "format", None)
    n_genes = int(X.shape[1])
    if fmt == "csc":
        w = np.diff(np.asarray(X.indptr, dtype=np.int64))
    elif fmt == "csr":
        w = np.bincount(np.asarray(X.indices), minlength=n_genes)[:n_genes]
    else:
        raise TypeError(f"sparse_gene_weights: expected a CSC or CSR matrix, got {type(X).__name__}")
    return w.astype(np.float64) + float(per_gene_cost)


def gene_ranges_for(X, world: int) -> list[tuple[int, int]]:
    """The ranges `asymptotic_wilcoxon_sharded` uses: by stored entries for scipy sparse input, by gene count otherwise."""
    if getattr(X, "format", None) in ("csc", "csr") and hasattr(X, "indptr"):
        return balanced_gene_ranges(sparse_gene_weights(X), world)
    return [rank_gene_range(int(X.shape[1]), r, world) for r in range(world)]


def gather_block_async(stage, recv_list, rank: int, world: int, dst: int = 0, group=None):
    """Start the gather of one contiguous block tensor to `dst`; returns the work handle.

    ``recv_list`` (on dst): one tensor per rank shaped like ``stage``; ignored elsewhere.
    """
    import torch.distributed as dist
    if stage.is_cuda and dist.get_backend(group) == "gloo":
        # test mode (no RCCL, e.g. several ranks sharing one GPU): stage the block through host memory
        cpu = stage.cpu()
        lst = [torch_empty_like_cpu(cpu) for _ in range(world)] if rank == dst else None
        dist.gather(cpu, gather_list=lst, dst=dst, group=group)
        if rank == dst:
            for r, t in zip(recv_list, lst):
                r.copy_(t)
        return _DoneWork()
    return dist.gather(stage, gather_list=recv_list if rank == dst else None, dst=dst, group=group, async_op=True)


def gather_block_p2p(stage, recv_bufs, rank: int, world: int, dst: int = 0, group=None):
    """Start the gather of one block whose WIDTH DIFFERS per rank (no padding to the widest rank): every source rank sends its
    ``stage`` (contiguous, [3, G, w_r]; nothing when w_r == 0), ``dst`` receives into ``recv_bufs[r]`` (one exactly-sized
    contiguous tensor per source rank r != dst, None where w_r == 0).  One grouped batch of sends / receives per block: over RCCL
    the 7 peers of an 8-GPU node push over 7 distinct xGMI links at once.  Returns the work handles."""
    import torch.distributed as dist
    if rank != dst and stage.numel() == 0:
        return []
    via_host = stage.is_cuda and dist.get_backend(group) == "gloo"  # test mode (no RCCL, e.g. several ranks sharing one GPU)
    if rank == dst:
        ops, hosts = [], []
        for r in range(world):
            if r == dst or recv_bufs[r] is None or recv_bufs[r].numel() == 0:
                continue
            buf = torch_empty_like_cpu(recv_bufs[r]) if via_host else recv_bufs[r]
            hosts.append((recv_bufs[r], buf))
            ops.append(dist.P2POp(dist.irecv, buf, r, group))
        works = dist.batch_isend_irecv(ops) if ops else []
        if via_host:
            for w in works:
                w.wait()
            for dev_buf, host_buf in hosts:
                dev_buf.copy_(host_buf)
            return []
        return works
    src = stage.cpu() if via_host else stage
    works = dist.batch_isend_irecv([dist.P2POp(dist.isend, src, dst, group)])
    if via_host:
        for w in works:
            w.wait()
        return []
    return works


class SharedHostPlanes:
    """ONE float64 [3][n_groups][n_genes] host result that every rank of the node maps (POSIX shared memory: a file under /dev/shm,
    i.e. what shm_open makes; unlinked as soon as every rank has mapped it, so nothing outlives the processes).  Rank `dst` creates it
    and broadcasts its name; every rank writes its own column range; after `barrier()` rank `dst` reads all of it.  ``array`` is backed
    by the mapping and keeps it alive: the DataFrame built over it (copy=False) needs no further bookkeeping.  Collective: every rank
    of `group` constructs it.  Without a process group (world 1) it is an anonymous mapping."""

    def __init__(self, n_groups: int, n_genes: int, group=None, dst: int = 0, directory: str | None = None):
        import torch.distributed as dist
        self.shape = (3, int(n_groups), int(n_genes))
        nbytes = max(8, 3 * int(n_groups) * int(n_genes) * 8)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        if self.world == 1:
            self._mm = mmap.mmap(-1, nbytes)
        else:
            directory = directory or ("/dev/shm" if os.path.isdir("/dev/shm") else None)
            box = [None]
            fd, err = -1, None
            if self.rank == dst:
                try:
                    if directory is None:
                        raise OSError("no /dev/shm on this host")
                    path = os.path.join(directory, f"illico_planes_{os.getpid()}_{uuid.uuid4().hex}")
                    fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                    os.ftruncate(fd, nbytes)
                    box[0] = path
                except OSError as e:  # every rank learns of it (no rank may be left waiting in a collective)
                    err, box[0] = e, f"!{e}"
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
            path = box[0]
            if path.startswith("!"):
                raise OSError(f"shared host planes: rank {dst} could not create the mapping: {path[1:]}") from err
            try:
                if self.rank != dst:
                    fd = os.open(path, os.O_RDWR)
                self._mm = mmap.mmap(fd, nbytes)
            finally:
                if fd >= 0:
                    os.close(fd)
                dist.barrier(group=group)      # every rank holds its mapping: the name can go
                if self.rank == dst:
                    try:
                        os.unlink(path)
                    except OSError:
                        pass
        self.array = np.frombuffer(self._mm, dtype=np.float64, count=3 * self.shape[1] * self.shape[2]).reshape(self.shape)

    def columns(self, lb: int, ub: int):
        """The three [G, ub - lb] windows of this rank's column range (row pitch = n_genes doubles: what `out_ld` of the C-ABI is for)."""
        return tuple(self.array[k][:, lb:ub] for k in range(3))

    def barrier(self):
        """Every rank's stores are in the mapping (a rank's engine call returns when its planes have landed)."""
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier(group=self.group)


class _DoneWork:
    def wait(self):
        return True


def torch_empty_like_cpu(t):
    import torch
    return torch.empty_like(t, device="cpu")


def asymptotic_wilcoxon_sharded(adata, is_log1p: bool, group_keys: str, reference=None, *, alternative="two-sided",
                                use_continuity=True, tie_correct=True, layer=None, n_blocks: int = 4, group=None,
                                compute_planes=None, column_loader=None, n_genes=None, var_names=None, groups=None,
                                gene_weights=None, tail: str = "host"):
    """Gene-sharded drop-in: rank 0 returns the DataFrame, the others None.

    Two ways to hand over the data:

    * every rank passes the same ``adata`` (the reference's argument, asymptotic_wilcoxon.py:71-83); a rank only ever
      touches the columns of its own range;
    * ``column_loader(lb, ub) -> X[:, lb:ub]`` (dense ndarray / device tensor / scipy CSC or CSR holding ONLY those
      columns) with ``n_genes``, ``groups`` (the per-cell labels, what ``adata.obs[group_keys]`` would hold) and, on
      rank 0, ``var_names``: a rank then never holds another rank's genes -- at BASELINE configs[4] the whole matrix is
      120 GB, a rank's shard 15 GB.  ``adata`` may be None.  ``gene_weights`` (one number per gene, e.g. stored entries,
      identical on every rank) balances the ranges; without it they are balanced by gene count.

    ``tail="host"`` (default): the result is ONE [3][G][M] array in shared host memory; every rank's engine writes its planes into
    its own column range of it over its own PCIe link (`SharedHostPlanes`; no data-path collective at all, like the reference's
    workers, asymptotic_wilcoxon.py:236-244).  ``tail="device"``: the planes are gathered over RCCL (exact-width p2p) into one device
    tensor on rank 0 and leave through one D2H -- for when rank 0's GPU is where they are wanted; 24 B per test into ONE link.

    ``compute_planes(X, grpc, lb, ub, **opts) -> (p, u, fc)`` defaults to the HIP engine of this rank's GPU
    (planes stay on the device until gathered over RCCL); tests inject a CPU function to cover the sharding
    and gather logic with the gloo backend.  With a loader, ``X`` is the loaded block and ``(lb, ub)`` are relative
    to it.
    """
    import pandas as pd
    import torch
    import torch.distributed as dist

    from illico_amd.utils.groups import encode_and_count_groups
    from illico_amd.utils.registry import data_handler_registry

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if column_loader is None:
        X = adata.layers[layer] if layer is not None else adata.X
        n_genes = int(X.shape[1])
        labels = np.asarray(adata.obs[group_keys])
        ranges = balanced_gene_ranges(gene_weights, world) if gene_weights is not None else gene_ranges_for(X, world)
        if rank == 0:
            var_names = np.asarray(adata.var_names)
    else:
        if n_genes is None or groups is None:
            raise ValueError("column_loader needs n_genes= and groups= (the per-cell labels)")
        X, n_genes, labels = None, int(n_genes), np.asarray(groups)
        ranges = (balanced_gene_ranges(gene_weights, world) if gene_weights is not None
                  else [rank_gene_range(n_genes, r, world) for r in range(world)])
        if rank == 0 and var_names is None:
            var_names = np.arange(n_genes).astype(str)
    if len(ranges) != world or ranges[0][0] != 0 or ranges[-1][1] != n_genes:
        raise ValueError("gene ranges do not cover the genes")
    unique, grpc = encode_and_count_groups(labels, reference)
    index_box: list = []
    index_thread = None
    if rank == 0:  # the result's row index does not depend on the statistics: built while the ranks compute (as in asymptotic_wilcoxon)
        from illico_amd.asymptotic_wilcoxon import _product_index
        cols = pd.Series(np.asarray(var_names), name="feature", dtype=str)
        rows = pd.Series(unique, name="pert", dtype=str)

        def build_index():
            try:
                index_box.append(_product_index(rows, cols))
            except BaseException as e:  # re-raised below
                index_box.append(e)

        index_thread = threading.Thread(target=build_index, name="illico-index", daemon=True)
        index_thread.start()
    G = int(grpc.counts.size)
    opts = dict(is_log1p=is_log1p, use_continuity=use_continuity, tie_correct=tie_correct, alternative=alternative)

    on_gpu = compute_planes is None
    if on_gpu:
        from illico_amd._lib import get_engine
        from illico_amd.utils.registry import KernelDataFormat
        eng = get_engine()
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_groups(grpc)
        dev = torch.device("cuda", eng.device)

        def compute_planes(Xb, grpc, lb, ub, out=None, **o):  # noqa: F811
            h = data_handler_registry.get(Xb)
            fmt = h.kernel_data_format()
            kw = dict(device_out=True) if out is None else dict(out=out)   # (out: host windows of the shared result)
            if fmt == KernelDataFormat.DENSE:
                return eng.run_dense(Xb, lb, ub, **kw, **o)
            return eng.run_sparse(fmt.value, Xb.data, Xb.indices, Xb.indptr, Xb.shape, lb, ub, **kw, **o)
    else:
        dev = torch.device("cpu")

    # every rank must issue the same number of gathers: blocks are defined per rank on its own range
    my_lb, my_ub = ranges[rank]
    if column_loader is not None:  # this rank's genes, and nothing else
        Xmine = column_loader(my_lb, my_ub) if my_ub > my_lb else None
        if Xmine is not None and int(Xmine.shape[1]) != my_ub - my_lb:
            raise ValueError(f"column_loader({my_lb}, {my_ub}) returned {Xmine.shape[1]} columns")
    if tail not in ("host", "device"):
        raise ValueError(f"tail must be 'host' or 'device', got {tail!r}")
    if tail == "host":
        # ---- every rank writes its own column range of one shared host result; nothing travels between GPUs ----
        shared = SharedHostPlanes(G, n_genes, group=group)
        try:
            for b in range(n_blocks if on_gpu else 1):  # (blocks: the engine's D2H of block b runs under the pass of block b + 1 inside the call)
                lb, ub = shard_bounds(my_ub - my_lb, n_blocks if on_gpu else 1)[b]
                if ub <= lb:
                    continue
                out = shared.columns(my_lb + lb, my_lb + ub)
                Xb, l0, u0 = (Xmine, lb, ub) if column_loader is not None else (X, my_lb + lb, my_lb + ub)
                if on_gpu:
                    compute_planes(Xb, grpc, l0, u0, out=out, **opts)
                else:  # an injected (CPU) function returns its planes
                    for dst_k, a in zip(out, compute_planes(Xb, grpc, l0, u0, **opts)):
                        dst_k[...] = a.numpy() if isinstance(a, torch.Tensor) else a
        finally:
            shared.barrier()  # (also on an error of this rank: the others must not wait for ever)
        if rank != 0:
            return None
        planes = shared.array
        index_thread.join()
        if isinstance(index_box[0], BaseException):
            raise index_box[0]
        return pd.DataFrame(
            {"p_value": planes[0].reshape(-1), "statistic": planes[1].reshape(-1), "fold_change": planes[2].reshape(-1)},
            index=index_box[0], copy=False)

    # ---- tail == "device" ----
    # Block b of rank r is exactly as wide as r's own b-th share (no padding to the widest rank: ranges balanced by stored entries differ
    # in width).  Rank 0 lays everything it computes and receives out in ONE device tensor [3][G][n_genes] -- its own blocks are
    # computed straight into it, a peer's block lands in an exactly-sized receive buffer and is copied into its column range on the
    # device -- and brings that tensor to the host ONCE, through the engine's pinned double buffer.
    widths_of = lambda r, b: (lambda se: se[1] - se[0])(shard_bounds(ranges[r][1] - ranges[r][0], n_blocks)[b])
    handles, keep, placed = [], [], []
    planes_dev = torch.empty((3, G, n_genes), dtype=torch.float64, device=dev) if rank == 0 else None
    for b in range(n_blocks):
        lb, ub = shard_bounds(my_ub - my_lb, n_blocks)[b]
        stage = torch.empty((3, G, ub - lb), dtype=torch.float64, device=dev)
        if ub > lb:
            if column_loader is not None:
                p, u, fc = compute_planes(Xmine, grpc, lb, ub, **opts)
            else:
                p, u, fc = compute_planes(X, grpc, my_lb + lb, my_lb + ub, **opts)
            for k, a in enumerate((p, u, fc)):
                stage[k] = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
            if rank == 0:
                planes_dev[:, :, my_lb + lb: my_lb + ub] = stage
        recv = None
        if rank == 0:
            recv = [None] * world
            for r in range(1, world):
                w = widths_of(r, b)
                if w > 0:
                    recv[r] = torch.empty((3, G, w), dtype=torch.float64, device=dev)
                    rl = ranges[r][0] + shard_bounds(ranges[r][1] - ranges[r][0], n_blocks)[b][0]
                    placed.append((recv[r], rl, rl + w))
        handles.extend(gather_block_p2p(stage, recv, rank, world, 0, group))
        keep.append((stage, recv))
    for h in handles:
        h.wait()
    if rank != 0:
        return None
    for buf, cl, cu in placed:
        planes_dev[:, :, cl:cu] = buf
    if on_gpu:
        planes = eng.planes_to_host(planes_dev)  # one device -> host copy of 24 bytes per test
    else:
        planes = planes_dev.numpy()
    index_thread.join()
    if isinstance(index_box[0], BaseException):
        raise index_box[0]
    return pd.DataFrame(
        {"p_value": planes[0].reshape(-1), "statistic": planes[1].reshape(-1), "fold_change": planes[2].reshape(-1)},
        index=index_box[0], copy=False)


def asymptotic_wilcoxon_threads(adata, is_log1p: bool, group_keys: str, reference=None, *, devices=None, alternative="two-sided",
                                use_continuity=True, tie_correct=True, layer=None, gene_weights=None, compute_planes=None):
    """The single-process multi-GPU form ("one process driving 8 devices with host threads"): one host thread and one
    engine context per entry of ``devices`` (default: every visible GPU; an id may repeat -- two contexts on one GPU), each computing one
    contiguous gene range of the HOST matrix ``adata.X`` straight into its column range of one [3][G][M] host result.  No process
    group, no collective: ctypes releases the GIL inside the engine calls, the reference's own model ("threads, never processes",
    README.md:6; asymptotic_wilcoxon.py:236-244).  Returns the reference's DataFrame.

    ``compute_planes(X, grpc, lb, ub, out, device_index, **opts)`` replaces the engine (tests)."""
    import pandas as pd

    from illico_amd.asymptotic_wilcoxon import _product_index
    from illico_amd.utils.groups import encode_and_count_groups
    from illico_amd.utils.registry import KernelDataFormat, data_handler_registry

    X = adata.layers[layer] if layer is not None else adata.X
    if alternative not in ("two-sided", "less", "greater"):
        raise ValueError(f"Unsupported alternative hypothesis: {alternative}")
    handler = data_handler_registry.get(X)  # KeyError for an unsupported container, like the reference (registry.py:58)
    fmt = handler.kernel_data_format()
    if devices is None:
        import torch
        devices = list(range(torch.cuda.device_count()))
    devices = list(devices)
    if not devices:
        raise RuntimeError("asymptotic_wilcoxon_threads: no GPU visible and no devices given (there is no CPU fallback)")
    world = len(devices)
    n_genes = int(X.shape[1])
    unique, grpc = encode_and_count_groups(adata.obs[group_keys], reference)
    G = int(grpc.counts.size)
    ranges = balanced_gene_ranges(gene_weights, world) if gene_weights is not None else gene_ranges_for(X, world)
    planes = np.empty((3, G, n_genes), dtype=np.float64)
    opts = dict(is_log1p=is_log1p, use_continuity=use_continuity, tie_correct=tie_correct, alternative=alternative)
    errors: list = [None] * world

    def work(i):
        lb, ub = ranges[i]
        if ub <= lb:
            return
        out = tuple(planes[k][:, lb:ub] for k in range(3))
        try:
            if compute_planes is not None:
                compute_planes(X, grpc, lb, ub, out, i, **opts)
                return
            from illico_amd._lib import Engine
            eng = Engine(devices[i])  # a context of this thread's own (a context is single-threaded, include/illico_hip.h)
            try:
                eng.set_groups(grpc)
                if fmt == KernelDataFormat.DENSE:
                    eng.run_dense(X, lb, ub, out=out, **opts)
                else:
                    eng.run_sparse(fmt.value, X.data, X.indices, X.indptr, X.shape, lb, ub, out=out, **opts)
            finally:
                eng.close()
        except BaseException as e:  # re-raised on the caller's thread
            errors[i] = e

    threads = [threading.Thread(target=work, args=(i,), name=f"illico-dev{devices[i]}-{i}") for i in range(world)]
    for t in threads:
        t.start()
    index = _product_index(pd.Series(unique, name="pert", dtype=str), pd.Series(np.asarray(adata.var_names), name="feature", dtype=str))
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return pd.DataFrame({"p_value": planes[0].reshape(-1), "statistic": planes[1].reshape(-1), "fold_change": planes[2].reshape(-1)},
                        index=index, copy=False)
