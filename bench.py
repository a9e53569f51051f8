#!/usr/bin/env python
"""Headline benchmark: (group x gene) tests/sec of the asymptotic Wilcoxon rank-sum hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5shard]

Workloads (BASELINE.json configs, synthetic data generated on the device, SURVEY.md 8d recipe: Poisson(gene mean
~U(0.1, 15)) counts as float32 with a fraction of the entries zeroed, one reference group of N/30 cells, the other
cells uniform over the remaining groups):
  c2       dense 300k cells x 8k genes x 2k groups, one-versus-reference (OVO), 50 % zeros   [default; configs[1]]
  c3       the same shape as CSC (float32 data, int32 indices), 90 % zeros, OVO                [configs[2]]
  c4       c2 with reference=None: one-versus-rest (OVR)                                      [configs[3]]
  c5shard  one GPU's gene shard of configs[4]: 1M cells x 3750 genes x 5k groups, dense OVO (8 ranks = the 30k genes)
A "step" is one pass of the hot path over the whole matrix, input resident in HBM, outputs (three float64 [G, M]
planes) left in HBM.

`--gpus N` with N > 1: when the process is not already a rank of a torch.distributed launch (the driver's
`python -m torch.distributed.run ... bench.py --gpus N`), it starts N ranks itself -- fresh child processes, one per
GPU, before this process has touched a GPU -- over RCCL (backend "nccl").  Genes shard across ranks; no input is ever
exchanged.  Default scaling is weak (every rank owns a full workload-sized gene shard); `--scaling strong` shards ONE
workload's genes by rank_gene_range.  A timed step ends, exactly as at N = 1, with the result planes resident in the HBM
of the GPU that computed them; the path's only collective -- the gather of the planes to rank 0 -- runs once after the
timed steps, is timed on its own and reported under "final_gather" (`--gather-in-step` issues it inside every step, per
gene block, overlapped with the next block's compute).

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline       achieved algorithmic GB/s of the dominant kernel (HIP events on the engine's stream, live, inside the
                 timed region) against the 8 TB/s HBM peak; `traffic` = PMC bytes per launch from profiles/traffic.json
  cpu_baseline   the CPU oracle (C restatement of illico's algorithm, oracle/, built -O3 -march=native on this box,
                 threads pinned one per physical core) on a bounded sample of the same workload: all physical cores, and
                 `at_8_threads` (the reference's headline setting, README.md:4)
  parity         genes of the LAST timed step's planes compared with the oracle: measured errors, not asserted ones
  timing_scopes  SURVEY.md 8d: (i) engine = ms_per_step, (ii) engine + H2D / D2H transfers, (iii) the drop-in call
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s

WORKLOADS = {
    "c2": dict(cells=300_000, genes=8_000, groups=2_000, sparsity=0.5, test="ovo", fmt="dense",
               label="dense {N}x{M}x{G} OVO (K562-shaped, BASELINE configs[1])"),
    "c3": dict(cells=300_000, genes=8_000, groups=2_000, sparsity=0.9, test="ovo", fmt="csc",
               label="CSC {N}x{M}x{G} OVO, 90% zeros (K562-shaped, BASELINE configs[2])"),
    "c4": dict(cells=300_000, genes=8_000, groups=2_000, sparsity=0.5, test="ovr", fmt="dense",
               label="dense {N}x{M}x{G} OVR (BASELINE configs[3])"),
    "c5shard": dict(cells=1_000_000, genes=3_750, groups=5_000, sparsity=0.5, test="ovo", fmt="dense",
                    label="dense {N}x{M}x{G} OVO: one GPU's gene shard of BASELINE configs[4] (1M x 30k x 5k over 8 GPUs)"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--cells", type=int, default=None)
    ap.add_argument("--genes", type=int, default=None, help="genes per GPU (weak scaling) / in total (--scaling strong)")
    ap.add_argument("--groups", type=int, default=None)
    ap.add_argument("--sparsity", type=float, default=None)
    ap.add_argument("--test", choices=["ovo", "ovr"], default=None)
    ap.add_argument("--format", choices=["dense", "csc", "csr"], default=None, dest="fmt")
    ap.add_argument("--values", choices=["counts", "continuous"], default="counts",
                    help="counts: the headline Poisson counts; continuous: log1p(counts * U(0.5,1.5)), the secondary stress of SURVEY.md 8d")
    ap.add_argument("--mean-max", type=float, default=15.0, help="gene means ~ U(0.1, mean-max); 15 = the reference's fixture (SURVEY.md 8d)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--gene-batch", type=int, default=0, help="genes per engine pass (0 = auto)")
    ap.add_argument("--gather-batches", type=int, default=8, help="gene blocks per gather (N>1)")
    ap.add_argument("--gather-in-step", action="store_true", help="N>1: gather every step's planes to rank 0 inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scopes", action="store_true", help="skip timing scopes (ii) and (iii)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-events", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-defer", action="store_true", help="dense passes wait for their route flags inside the call (no ILLICO_FLAG_DEFER)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of each CPU baseline run")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)       # test hook: gloo
    ap.add_argument("--share-device", action="store_true", help=argparse.SUPPRESS)  # test hook: every rank on GPU 0
    ap.add_argument("--engine-option", action="append", default=[], help="key=value passed to illico_ctx_set_option")
    args = ap.parse_args(argv)
    w = WORKLOADS[args.workload]
    for k in ("cells", "genes", "groups", "sparsity", "test", "fmt"):
        if getattr(args, k) is None:
            setattr(args, k, w[k])
    return args


# ---- synthetic workload (shared with tests/ and tools/) ---------------------------------------------------------------
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


def make_matrix(torch, n_cells, n_genes, sparsity, seed, device, continuous=False, mean_max=15.0):
    """Poisson(gene mean ~ U(0.1, 15)) float32 with `sparsity` of the entries zeroed, generated on device."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    # the row pitch is padded to a multiple of 128 bytes (the C-ABI takes any leading dimension): the fused kernels read
    # 256-byte row segments, which then cover two 128-byte lines instead of straddling three (C5 shard, 3750 genes: a pitch
    # of 15 000 bytes would misalign every row; C2's 32 000 bytes are aligned as they are)
    ld = (n_genes + 31) // 32 * 32
    X = torch.empty((n_cells, ld), dtype=torch.float32, device=device)[:, :n_genes]
    means = torch.empty(n_genes, device=device).uniform_(0.1, mean_max, generator=gen)
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


def compress(torch, X, fmt):
    """Device-resident CSC / CSR arrays (float32 data, int32 indices / indptr) of a dense device matrix, block by block."""
    dev = X.device
    N, M = X.shape
    datas, idxs, cnts = [], [], []
    if fmt == "csc":  # CSC = CSR of X^T
        for j0 in range(0, M, 256):
            Xb = X[:, j0:j0 + 256].t().contiguous()
            nz = Xb != 0
            cnts.append(nz.sum(1)); idxs.append(nz.nonzero()[:, 1].to(torch.int32)); datas.append(Xb[nz])
    else:
        for r0 in range(0, N, 16384):
            Xb = X[r0:r0 + 16384]
            nz = Xb != 0
            cnts.append(nz.sum(1)); idxs.append(nz.nonzero()[:, 1].to(torch.int32)); datas.append(Xb[nz])
    data, indices = torch.cat(datas).contiguous(), torch.cat(idxs).contiguous()
    indptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cat(cnts).cumsum(0)]).to(torch.int32).contiguous()
    return data, indices, indptr


# ---- N > 1 without a launcher: start the ranks ourselves ----------------------------------------------------------------
def spawn_ranks(args) -> int:
    """Fresh child processes, one per rank; this parent never initialises a GPU (device_count() does not)."""
    import socket
    import torch
    n = args.gpus
    have = torch.cuda.device_count()
    if not args.share_device and have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node; refusing to measure fewer ranks than asked")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but the launcher started {world} rank(s); measuring {world}", file=sys.stderr)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    if args.share_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    from illico_amd._lib import Engine
    from illico_amd.distributed import gather_block_async, rank_gene_range, shard_bounds

    N, G = args.cells, args.groups
    if args.scaling == "strong":  # one workload's genes, sharded
        g_lb, g_ub = rank_gene_range(args.genes, rank, world)
        M, M_total = g_ub - g_lb, args.genes
    else:
        M, M_total = args.genes, args.genes * world
    ovr = args.test == "ovr"
    sparse_fmt = args.fmt if args.fmt != "dense" else None
    codes = make_labels(N, G, args.seed)
    grpc = group_container(codes, G, ovr)
    X = make_matrix(torch, N, M, args.sparsity, args.seed + 1000 * rank, device, args.values == "continuous", args.mean_max)  # this rank's gene shard
    csx, nnz = None, None
    if sparse_fmt:
        csx = compress(torch, X, sparse_fmt)
        nnz = int(csx[0].numel())
        del X
        X = None
        torch.cuda.empty_cache()
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
    blocks = [b for b in shard_bounds(M, n_blocks) if b[1] > b[0]]
    # one contiguous (3, G, w) staging tensor per gene block: the engine writes its planes straight into it.  Two sets, used
    # by alternate steps (a consumer reads step k's planes while step k + 1 computes): with ILLICO_FLAG_DEFER a dense pass is
    # then enqueued before the previous one's route flags have been looked at -- no host round trip between passes.
    stage_sets = [[torch.empty((3, G, ub - lb), dtype=torch.float64, device=device) for (lb, ub) in blocks] for _ in range(2)]
    stages = stage_sets[0]
    step_no = [0]
    recvs = None
    if world > 1 and rank == 0:
        recvs = [[torch.empty_like(st) for _ in range(world)] for st in stages]

    def run_block(lb, ub, out):
        if sparse_fmt:
            eng.run_sparse(sparse_fmt, csx[0], csx[1], csx[2], (N, M), lb, ub, out=out)
        else:
            eng.run_dense(X, lb, ub, out=out, defer=not in_step and not args.no_defer)

    def step():
        nonlocal stages
        stages = stage_sets[step_no[0] & 1]
        step_no[0] += 1
        handles = []
        for b, (lb, ub) in enumerate(blocks):
            st = stages[b]
            run_block(lb, ub, (st[0], st[1], st[2]))
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
        eng.synchronize()  # completes a deferred pass (genes the fused route could not take), then waits for the stream
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # first call of a fresh context: scratch allocation, route decision -- what a single drop-in call pays
    sync()
    t0 = time.perf_counter()
    step()
    eng.synchronize()
    first_call_ms = (time.perf_counter() - t0) * 1e3
    # The first ~8 passes of a fresh process run up to 6 % slower than the steady state (tools/ramp.py): a few settling
    # passes before the W warm-up steps, so that a small W still measures the steady state.  Untimed, like the warm-up.
    settle = 5
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
    tests_per_step = G * M_total
    value = tests_per_step / (dt / args.steps)

    # what this box's HBM delivers to a plain streaming read (boxes of the pool differ by up to 15 %: 1.65 vs 1.90 ms for the
    # same C2 kernel): torch's sum over the resident input
    calib = None
    if rank == 0:
        src = X if X is not None else csx[0]
        src.sum(); torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(5):
            src.sum()
        torch.cuda.synchronize()
        calib = {"stream_read_GBs": round(src.numel() * src.element_size() * 5 / (time.perf_counter() - tc) / 1e9, 1),
                 "what": "torch .sum() over the resident input, 5 passes: the box's own streaming-read rate, for comparing runs on different boxes"}

    if rank == 0:
        # ---- roofline of the dominant kernel (HIP events recorded on the engine's stream) ----
        # input once + 4 B per cell of codes + three f64 planes
        if sparse_fmt:
            alg_bytes_step = nnz * 8 + ((M if sparse_fmt == "csc" else N) + 1) * 4 + 4 * N + 24 * G * M
        else:
            alg_bytes_step = N * M * 4 + 4 * N + 24 * G * M
        wl_key = {"workload": args.workload, "cells": N, "genes_per_gpu": M, "groups": G, "test": args.test, "format": args.fmt,
                  "values": args.values, "sparsity": args.sparsity}
        if args.mean_max != 15.0:
            wl_key["mean_max"] = args.mean_max
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
                    for ent in json.loads(tf.read_text()).get("entries", []):
                        if ent.get("kernel_id") == dom and all(ent.get("workload", {}).get(k) == v for k, v in wl_key.items()) \
                                and launches == args.steps * ent.get("launches_per_step", 1):
                            traffic = ent.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches / args.steps,
                        "algorithmic_bytes_per_launch": int(bytes_per_launch),
                        "all_kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in breakdown.items()},
                        "all_kernels_note": "one untimed step with events around every kernel, taken before the timed region",
                        "pipeline_achieved": round(alg_bytes_step / (ms_per_step * 1e-3) / 1e9, 2),
                        "pipeline_frac": round(alg_bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

        def host_columns(cols):
            """Dense float32 host copy of a few of this rank's genes (whatever the input format)."""
            if not sparse_fmt:
                return X[:, cols].contiguous().cpu().numpy()
            if sparse_fmt == "csc":
                out = np.zeros((N, len(cols)), dtype=np.float32)
                ip = csx[2].cpu().numpy()
                for i, c in enumerate(cols):
                    s, e = int(ip[c]), int(ip[c + 1])
                    out[csx[1][s:e].cpu().numpy(), i] = csx[0][s:e].cpu().numpy()
                return out
            raise SystemExit("parity / CPU baseline sampling for device CSR input is not implemented in bench.py")

        # ---- parity of what was just timed: a few genes of the final step's planes against the oracle ----
        parity = None
        if not args.no_parity and sparse_fmt != "csr":
            import oracle
            cols = sorted({0, M // 3, M // 2, M - 1})
            want = oracle.run(host_columns(cols), grpc, batch_size=1, n_threads=min(len(cols), 8))
            got = [np.empty((G, len(cols))) for _ in range(3)]
            for i, c in enumerate(cols):
                for b, (lb, ub) in enumerate(blocks):
                    if lb <= c < ub:
                        for k in range(3):
                            got[k][:, i] = stages[b][k][:, c - lb].cpu().numpy()
            mask = np.ones(G, dtype=bool)
            if not ovr:
                mask[0] = False  # the reference leaves the reference group's row unspecified
            with np.errstate(divide="ignore", invalid="ignore"):
                perr = np.abs(got[0][mask] - want[0][mask]) / np.abs(want[0][mask])
                ferr = np.abs(got[2] - want[2]) / np.abs(want[2])
            perr = np.where(got[0][mask] == want[0][mask], 0.0, perr)
            ferr = np.where((got[2] == want[2]) | (np.isnan(got[2]) & np.isnan(want[2])), 0.0, ferr)
            parity = {"genes_checked": cols, "tests_checked": int(mask.sum() * len(cols)),
                      "statistic_mismatches": int((got[1][mask] != want[1][mask]).sum()),
                      "p_value_max_rel_err": float(np.max(perr)), "fold_change_max_rel_err": float(np.max(ferr)),
                      "bar": "statistic exact; p_value and fold_change rtol 1e-12", "against": "oracle/ (CPU restatement pinned to the reference's outputs)"}

        # ---- CPU baseline: the oracle on this box's host cores, bounded sample, threads pinned ----
        cpu = None
        if not args.no_cpu_baseline and world == 1 and sparse_fmt != "csr":  # contract: rank 0 at N = 1 only
            import oracle
            oracle.use_native(True)
            n_phys = oracle.pin_threads(True)
            from scipy import sparse as sp

            def sample(ns):
                if sparse_fmt == "csc":
                    ip = csx[2][: ns + 1].cpu().numpy().astype(np.int64)
                    return sp.csc_matrix((csx[0][: ip[-1]].cpu().numpy(), csx[1][: ip[-1]].cpu().numpy(), ip), shape=(N, ns))
                return X[:, :ns].contiguous().cpu().numpy()

            def timed(n_threads, batch, seconds):
                """Probe with one batch per thread, then a run sized to ~`seconds`; returns (tests/s, genes, wall)."""
                ns = min(M, n_threads * batch)
                Xs = sample(ns)
                t1 = time.perf_counter()
                oracle.run(Xs, grpc, batch_size=batch, n_threads=n_threads)
                el = time.perf_counter() - t1
                ns2 = int(min(M, (G * ns / el) * seconds / G))
                ns2 = max(n_threads * batch, (ns2 // (n_threads * batch)) * (n_threads * batch))
                if ns2 > ns:
                    ns = min(ns2, M)
                    Xs = sample(ns)
                    t1 = time.perf_counter()
                    oracle.run(Xs, grpc, batch_size=batch, n_threads=n_threads)
                    el = time.perf_counter() - t1
                return G * ns / el, ns, el

            # all physical cores: chunks wide enough that the row gathers use whole cache lines, narrow enough to keep every
            # core busy within the sample; 8 threads: the reference's 256-gene chunks (README.md:124 benchmarks)
            b_all = int(max(8, min(256, M // max(n_phys, 1))))
            v_all, ns_all, el_all = timed(n_phys, b_all, args.cpu_seconds)
            v_8, ns_8, el_8 = timed(min(8, n_phys), int(min(256, max(8, M // 8))), args.cpu_seconds)
            oracle.pin_threads(False)
            oracle.use_native(False)
            what = f"same {N}x{M}x{G} {args.test.upper()} {args.fmt} workload; oracle/ (C restatement of illico's algorithm), -O3 -march=native, OpenMP over gene chunks, threads pinned one per physical core"
            cpu = {"value": round(v_all, 1), "unit": "tests/s", "cores": n_phys, "kind": "port",
                   "sample": f"first {ns_all} genes in chunks of {b_all}, {el_all:.1f}s wall; {what}",
                   "at_8_threads": {"value": round(v_8, 1), "cores": min(8, n_phys), "sample": f"first {ns_8} genes, {el_8:.1f}s wall",
                                    "note": "the reference's headline setting (README.md:4: 8 threads)"},
                   "logical_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}

        # ---- timing scopes (ii) engine + transfers and (iii) the drop-in call, SURVEY.md 8d ----
        scopes = {"engine_ms": round(ms_per_step, 4), "first_call_ms": round(first_call_ms, 3)}
        if not args.no_scopes and world == 1:
            try:
                import pandas as pd
                from illico_amd import AnnDataLite, asymptotic_wilcoxon
                if sparse_fmt:
                    from scipy import sparse as sp
                    ctor = sp.csc_matrix if sparse_fmt == "csc" else sp.csr_matrix
                    Xh = ctor((csx[0].cpu().numpy(), csx[1].cpu().numpy(), csx[2].cpu().numpy()), shape=(N, M))
                    run_host = lambda: eng.run_sparse(sparse_fmt, Xh.data, Xh.indices, Xh.indptr, (N, M), 0, M)
                else:
                    Xh = X.contiguous().cpu().numpy()
                    run_host = lambda: eng.run_dense(Xh, 0, M)
                run_host()  # scratch for the staged route
                t1 = time.perf_counter()
                run_host()
                scopes["engine_plus_transfers_ms"] = round((time.perf_counter() - t1) * 1e3, 2)
                labels = np.where(codes == 0, "non-targeting", np.char.add("pert_", np.char.zfill(codes.astype(str), 5)))
                adata = AnnDataLite(Xh, obs=pd.DataFrame({"pert": labels}))
                kw = dict(is_log1p=False, group_keys="pert", reference=None if ovr else "non-targeting")
                t1 = time.perf_counter()
                df = asymptotic_wilcoxon(adata, **kw)
                scopes["drop_in_call_ms"] = round((time.perf_counter() - t1) * 1e3, 2)
                scopes["drop_in_rows"] = int(len(df))
                scopes["note"] = ("(ii) host-resident input -> host planes: pageable H2D of the matrix and D2H of 24 B per test included; "
                                  "(iii) illico_amd.asymptotic_wilcoxon(adata, ...) on the same host matrix: group encoding, (ii), DataFrame assembly")
                del df, adata, Xh
            except MemoryError as e:  # a host too small for a second copy of the workload
                scopes["skipped"] = f"host memory: {e}"

        result = {
            "metric": "(group x gene) tests/sec", "value": round(value, 1), "unit": "tests/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload]["label"].format(N=N, M=M, G=G) + ("" if args.values == "counts" else " [continuous values]"),
                       "workload_id": args.workload, "cells": N, "genes_per_gpu": M, "genes_total": M_total, "groups": G, "format": args.fmt,
                       "test": args.test, "sparsity": args.sparsity, "gene_mean_max": args.mean_max, "nnz_per_gpu": nnz, "settle_steps": settle,
                       "parallelism": f"gene-shard x{world}" if world > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "timing_scopes": scopes, "box_calibration": calib,
        }
        if world > 1:
            plane_bytes = 24 * G * M
            result["final_gather"] = ({"in_timed_step": True, "blocks_per_step": len(blocks), "bytes_per_rank_per_step": plane_bytes}
                                      if in_step else
                                      {"in_timed_step": False, "ms": round(gather_ms, 3), "bytes_per_rank": plane_bytes,
                                       "bytes_into_rank0": plane_bytes * (world - 1),
                                       "one_pass_plus_gather_ms": round(ms_per_step + gather_ms, 3),
                                       "tests_per_s_pass_plus_gather": round(tests_per_step / ((ms_per_step + gather_ms) * 1e-3), 1),
                                       "note": "timed steps end with each rank's planes in its own HBM (as at N=1); the one "
                                               "collective of the path runs once per job, after them"})
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
