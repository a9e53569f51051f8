#!/usr/bin/env python
"""Headline benchmark: (group x gene) tests/sec of the asymptotic Wilcoxon rank-sum hot path.

Workload (BASELINE.json configs[1], "C2"): synthetic dense float32 300k cells x 8k genes x 2k groups,
one-versus-reference (OVO), K562-shaped: Poisson(gene mean ~U(0.1,15)) counts with 50% zeros, one
reference group of N/30 cells, the other cells uniform over the remaining groups.
A "step" is one pass of the hot path over the whole matrix, input resident in HBM, outputs (three
float64 [G, M] planes) left in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
For N > 1 the driver launches one rank per GPU with torch.distributed.run; genes shard across ranks
(each rank owns a full C2-sized gene shard: weak scaling) and no input is ever exchanged.  A timed step
ends, exactly as at N = 1, with the result planes resident in the HBM of the GPU that computed them.
The path's only collective -- the final gather of the planes to rank 0 over RCCL/xGMI -- runs ONCE after
the timed steps, is timed on its own and reported under "final_gather" (it moves 24 B per test:
384 MB per rank at C2).  `--gather-in-step` instead issues the gather inside every step, per gene block,
overlapped with the next block's compute.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      achieved algorithmic GB/s of the dominant kernel (HIP events, live) vs the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (C port of illico's algorithm, oracle/) timed on this box's host cores on a
                bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=300_000)
    ap.add_argument("--genes", type=int, default=8_000, help="genes per GPU (weak scaling)")
    ap.add_argument("--groups", type=int, default=2_000)
    ap.add_argument("--sparsity", type=float, default=0.5)
    ap.add_argument("--test", choices=["ovo", "ovr"], default="ovo")
    ap.add_argument("--values", choices=["counts", "continuous"], default="counts",
                    help="counts: the headline Poisson counts; continuous: log1p(counts * U(0.5,1.5)), the secondary stress of SURVEY.md 8d")
    ap.add_argument("--gene-batch", type=int, default=0, help="genes per engine pass (0 = auto)")
    ap.add_argument("--gather-batches", type=int, default=8, help="gene blocks per gather (N>1)")
    ap.add_argument("--gather-in-step", action="store_true", help="N>1: gather every step's planes to rank 0 inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)       # test hook: gloo
    ap.add_argument("--share-device", action="store_true", help=argparse.SUPPRESS)  # test hook: every rank on GPU 0
    ap.add_argument("--engine-option", action="append", default=[], help="key=value passed to illico_ctx_set_option")
    return ap.parse_args()


def make_labels(n_cells, n_groups, seed):
    """One reference group 'non-targeting' of round(N/30) cells, the rest uniform over G-1 labels, shuffled."""
    rng = np.random.RandomState(seed)
    n_ref = max(1, int(round(n_cells / 30)))
    codes = np.concatenate([np.zeros(n_ref, dtype=np.int64), 1 + rng.randint(0, n_groups - 1, size=n_cells - n_ref)])
    rng.shuffle(codes)
    return codes  # code 0 = reference; code order == label order ("non-targeting" < "pert_%05d")


def group_container(codes, n_groups, ovr):
    from illico_amd.utils.groups import GroupContainer
    counts = np.bincount(codes, minlength=n_groups).astype(np.int64)
    indices = np.argsort(codes, kind="stable").astype(np.int64)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return GroupContainer(codes.astype(np.int64), counts, indices, indptr, -1 if ovr else 0)


def make_matrix(torch, n_cells, n_genes, sparsity, seed, device, continuous=False):
    """Poisson(gene mean ~ U(0.1, 15)) float32 with `sparsity` of the entries zeroed, generated on device."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    X = torch.empty((n_cells, n_genes), dtype=torch.float32, device=device)
    means = torch.empty(n_genes, device=device).uniform_(0.1, 15.0, generator=gen)
    step = 256
    for j in range(0, n_genes, step):
        m = means[j:j + step]
        rates = m.unsqueeze(0).expand(n_cells, m.numel()).contiguous()
        blk = torch.poisson(rates, generator=gen)
        keep = torch.rand(blk.shape, device=device, generator=gen) >= sparsity
        if continuous:  # normalised-like data: (almost) no ties among the non-zeros
            blk = torch.log1p(blk * torch.empty_like(blk).uniform_(0.5, 1.5, generator=gen))
        X[:, j:j + step] = blk * keep
    return X


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    from illico_amd._lib import Engine
    from illico_amd.distributed import gather_block_async, shard_bounds

    N, M, G = args.cells, args.genes, args.groups
    ovr = args.test == "ovr"
    codes = make_labels(N, G, args.seed)
    grpc = group_container(codes, G, ovr)
    X = make_matrix(torch, N, M, args.sparsity, args.seed + 1000 * rank, device, args.values == "continuous")  # this rank's gene shard
    torch.cuda.synchronize()

    eng = Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.gene_batch:
        eng.set_option("gene_batch", args.gene_batch)
    for kv in args.engine_option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.set_groups(grpc)
    in_step = world > 1 and args.gather_in_step
    n_blocks = max(1, args.gather_batches) if in_step else 1
    blocks = shard_bounds(M, n_blocks)
    # one contiguous (3, G, w) staging tensor per gene block: the engine writes its planes straight into it
    stages = [torch.empty((3, G, ub - lb), dtype=torch.float64, device=device) for (lb, ub) in blocks]
    recvs = None
    if world > 1 and rank == 0:
        recvs = [[torch.empty_like(st) for _ in range(world)] for st in stages]

    def step():
        handles = []
        for b, (lb, ub) in enumerate(blocks):
            st = stages[b]
            eng.run_dense(X, lb, ub, out=(st[0], st[1], st[2]))
            if in_step:
                handles.append(gather_block_async(st, recvs[b] if rank == 0 else None, rank, world))
        for h in handles:
            h.wait()

    def final_gather():
        """The path's one collective: every rank's planes to rank 0 (torch.distributed.gather over RCCL)."""
        hs = [gather_block_async(stages[b], recvs[b] if rank == 0 else None, rank, world) for b in range(len(blocks))]
        for h in hs:
            h.wait()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The first ~8 passes of a fresh process run up to 6 % slower than the steady state (tools/ramp.py: 2.13, 2.09, 2.07,
    # 2.05, 2.04, 2.03, 2.02 ... 2.01 ms): a few settling passes before the W warm-up steps, so that a small W still
    # measures the steady state.  Untimed, like the warm-up and the data generation.
    settle = 6
    for _ in range(settle + args.warmup):
        step()
    sync()
    # One untimed step with HIP events around every kernel: the per-kernel breakdown, and which kernel dominates.
    # Inside the timed region only that kernel carries events (each event pair drains the stream around a launch).
    eng.profile(True)
    eng.profile_reset()
    step()
    sync()
    breakdown = eng.profile_get()
    dom = max(breakdown.items(), key=lambda kv: kv[1]["ms"])[0] if breakdown else None
    eng.profile_only(dom)
    eng.profile_reset()
    if args.no_events:  # diagnostic only: the contract wants the dominant kernel timed inside the timed region
        eng.profile(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    prof = eng.profile_get()
    eng.profile(False)
    eng.profile_only(None)
    gather_ms = None
    if world > 1:
        red_dev = device if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if not in_step:  # the final gather, once, timed on its own (second call: RCCL connections already set up)
            final_gather()
            sync()
            tg = time.perf_counter()
            final_gather()
            sync()
            t = torch.tensor([time.perf_counter() - tg], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gather_ms = float(t.item()) * 1e3

    ms_per_step = dt / args.steps * 1e3
    tests_per_step = G * M * world
    value = tests_per_step / (dt / args.steps)

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events recorded on the engine's stream) ----
        alg_bytes_step = N * M * 4 + 4 * N + 24 * G * M  # input once + codes + three f64 planes
        roofline = None
        if dom and dom in prof:
            launches = prof[dom]["launches"]
            avg_ms = prof[dom]["ms"] / launches
            bytes_per_launch = alg_bytes_step * args.steps / launches
            achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
            traffic = None
            tf = ROOT / "profiles" / "traffic.json"
            if tf.exists():
                try:  # PMC traffic is only valid for the workload it was collected on
                    ent = json.loads(tf.read_text()).get(dom, {})
                    w = ent.get("workload", {})
                    if (w.get("cells"), w.get("genes_per_gpu"), w.get("groups"), w.get("test"), w.get("values")) == \
                            (N, M, G, args.test, args.values) and launches == args.steps * w.get("launches_per_step", 1):
                        traffic = ent.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches / args.steps,
                        "algorithmic_bytes_per_launch": int(bytes_per_launch),
                        "all_kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in breakdown.items()},
                        "all_kernels_note": "one untimed step with events around every kernel, taken before the timed region",
                        "pipeline_achieved": round(alg_bytes_step / (ms_per_step * 1e-3) / 1e9, 2)}

        # ---- CPU baseline: the oracle (C port of illico's algorithm) on this box's host cores ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # contract: CPU baseline on rank 0 at N = 1 only
            import oracle
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            ns = min(M, 32)
            Xs = X[:, :ns].contiguous().cpu().numpy()
            t1 = time.perf_counter()
            oracle.run(Xs, grpc, batch_size=max(1, -(-ns // cores)), n_threads=cores)
            el = time.perf_counter() - t1
            rate = G * ns / el
            ns2 = int(min(M, max(ns, rate * args.cpu_seconds / G)))
            ns2 = max(cores, (ns2 // cores) * cores)
            if ns2 > ns * 2:
                Xs = X[:, :ns2].contiguous().cpu().numpy()
                t1 = time.perf_counter()
                oracle.run(Xs, grpc, batch_size=min(256, max(1, -(-ns2 // cores))), n_threads=cores)
                el = time.perf_counter() - t1
                ns = ns2
            cpu = {"value": round(G * ns / el, 1), "unit": "tests/s", "cores": cores, "kind": "port",
                   "sample": f"first {ns} genes of the same {N}x{M}x{G} {args.test.upper()} workload, {el:.1f}s wall, "
                             f"oracle/ (C restatement of illico's algorithm), OpenMP over gene chunks"}

        result = {
            "metric": "(group x gene) tests/sec", "value": round(value, 1), "unit": "tests/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"dense {N}x{M}x{G} {args.test.upper()} (K562-shaped, BASELINE configs[1])" if not ovr
                                    else f"dense {N}x{M}x{G} OVR") + ("" if args.values == "counts" else " [continuous values]"), "cells": N, "genes_per_gpu": M, "groups": G,
                       "sparsity": args.sparsity, "settle_steps": settle, "parallelism": f"gene-shard x{world}" if world > 1 else "single GPU",
                       "p_value_rtol_vs_cpu": 1e-12},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if world > 1:
            plane_bytes = 24 * G * M
            result["final_gather"] = ({"in_timed_step": True, "blocks_per_step": len(blocks), "bytes_per_rank_per_step": plane_bytes}
                                      if in_step else
                                      {"in_timed_step": False, "ms": round(gather_ms, 3), "bytes_per_rank": plane_bytes,
                                       "bytes_into_rank0": plane_bytes * (world - 1),
                                       "one_pass_plus_gather_ms": round(ms_per_step + gather_ms, 3),
                                       "note": "timed steps end with each rank's planes in its own HBM (as at N=1); the one "
                                               "collective of the path runs once per job, after them"})
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
