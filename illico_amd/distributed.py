"""Gene sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

Genes are independent units (every reference kernel takes an arbitrary contiguous gene range,
illico/asymptotic_wilcoxon.py:213-241), so rank r of R computes genes [r*M/R, (r+1)*M/R) with no input
exchange; the only collective is the gather of the three float64 result planes to rank 0, issued per
gene block so that it overlaps the next block's compute.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, parts: int) -> list[tuple[int, int]]:
    """Contiguous, balanced (+-1) split of range(n) into EXACTLY `parts` windows (empty ones when n < parts: every
    rank must issue the same number of gathers whatever its own width)."""
    parts = max(1, int(parts))
    edges = [(n * i) // parts for i in range(parts + 1)]
    return [(edges[i], edges[i + 1]) for i in range(parts)]


def rank_gene_range(n_genes: int, rank: int, world: int) -> tuple[int, int]:
    return ((n_genes * rank) // world, (n_genes * (rank + 1)) // world)


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


class _DoneWork:
    def wait(self):
        return True


def torch_empty_like_cpu(t):
    import torch
    return torch.empty_like(t, device="cpu")


def asymptotic_wilcoxon_sharded(adata, is_log1p: bool, group_keys: str, reference=None, *, alternative="two-sided",
                                use_continuity=True, tie_correct=True, layer=None, n_blocks: int = 4, group=None,
                                compute_planes=None):
    """Gene-sharded drop-in: every rank passes the same ``adata``; rank 0 returns the DataFrame, the others None.

    ``compute_planes(X, grpc, lb, ub, **opts) -> (p, u, fc)`` defaults to the HIP engine of this rank's GPU
    (planes stay on the device until gathered over RCCL); tests inject a CPU function to cover the sharding
    and gather logic with the gloo backend.
    """
    import pandas as pd
    import torch
    import torch.distributed as dist

    from illico_amd.utils.groups import encode_and_count_groups
    from illico_amd.utils.registry import data_handler_registry

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    X = adata.layers[layer] if layer is not None else adata.X
    handler = data_handler_registry.get(X)
    unique, grpc = encode_and_count_groups(np.asarray(adata.obs[group_keys]), reference)
    n_genes, G = X.shape[1], int(grpc.counts.size)
    opts = dict(is_log1p=is_log1p, use_continuity=use_continuity, tie_correct=tie_correct, alternative=alternative)

    on_gpu = compute_planes is None
    if on_gpu:
        from illico_amd._lib import get_engine
        from illico_amd.utils.registry import KernelDataFormat
        eng = get_engine()
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_groups(grpc)
        fmt = handler.kernel_data_format()
        dev = torch.device("cuda", eng.device)

        def compute_planes(X, grpc, lb, ub, **o):  # noqa: F811
            if fmt == KernelDataFormat.DENSE:
                return eng.run_dense(X, lb, ub, device_out=True, **o)
            return eng.run_sparse(fmt.value, X.data, X.indices, X.indptr, X.shape, lb, ub, device_out=True, **o)
    else:
        dev = torch.device("cpu")

    # every rank must issue the same number of gathers: blocks are defined per rank on its own range
    ranges = [rank_gene_range(n_genes, r, world) for r in range(world)]
    my_lb, my_ub = ranges[rank]
    handles, stages, recvs = [], [], []
    for b in range(n_blocks):
        lb, ub = shard_bounds(my_ub - my_lb, n_blocks)[b] if my_ub > my_lb else (0, 0)
        lb, ub = my_lb + lb, my_lb + ub
        widths = []
        for r in range(world):
            rl, ru = ranges[r]
            bl, bu = shard_bounds(ru - rl, n_blocks)[b] if ru > rl else (0, 0)
            widths.append(bu - bl)
        wmax = max(widths)
        stage = torch.zeros((3, G, wmax), dtype=torch.float64, device=dev)
        if ub > lb:
            p, u, fc = compute_planes(X, grpc, lb, ub, **opts)
            for k, a in enumerate((p, u, fc)):
                stage[k, :, : ub - lb] = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        recv = [torch.empty_like(stage) for _ in range(world)] if rank == 0 else None
        handles.append(gather_block_async(stage, recv, rank, world, 0, group))
        stages.append(stage)
        recvs.append((recv, widths))
    for h in handles:
        h.wait()
    if rank != 0:
        return None
    planes = np.empty((3, G, n_genes), dtype=np.float64)
    for b, (recv, widths) in enumerate(recvs):
        for r in range(world):
            rl, ru = ranges[r]
            bl, bu = shard_bounds(ru - rl, n_blocks)[b] if ru > rl else (0, 0)
            if bu > bl:
                planes[:, :, rl + bl: rl + bu] = recv[r][:, :, : bu - bl].cpu().numpy()
    cols = pd.Series(np.asarray(adata.var_names), name="feature", dtype=str)
    rows = pd.Series(unique, name="pert", dtype=str)
    return pd.DataFrame(
        {"p_value": planes[0].reshape(-1), "statistic": planes[1].reshape(-1), "fold_change": planes[2].reshape(-1)},
        index=pd.MultiIndex.from_product([rows, cols], names=["pert", "feature"]), copy=False)
