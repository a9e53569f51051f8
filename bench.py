#!/usr/bin/env python
"""Headline benchmark: (group x gene) tests/sec of the asymptotic Wilcoxon rank-sum hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5shard|c5]

Workloads (BASELINE.json configs, synthetic data generated on the device, SURVEY.md 8d recipe: Poisson(gene mean
~U(0.1, 15)) counts as float32 with a fraction of the entries zeroed, one reference group of N/30 cells, the other
cells uniform over the remaining groups):
  c2       dense 300k cells x 8k genes x 2k groups, one-versus-reference (OVO), 50 % zeros   [default; configs[1]]
  c3       the same shape as CSC (float32 data, int32 indices), 90 % zeros, OVO                [configs[2]]
  c4       c2 with reference=None: one-versus-rest (OVR)                                      [configs[3]]
  c5       dense 1M cells x 30k genes x 5k groups, OVO: configs[4] whole (120 GB: fits one 288 GB MI355X)
  c5shard  one GPU's gene shard of configs[4] as a workload of its own: 1M cells x 3750 genes x 5k groups
A "step" is one pass of the hot path over the workload's whole matrix, input resident in HBM.  At N = 1 the three
float64 [G, M] result planes stay in HBM.

`--gpus N` with N > 1 (started by the driver's `python -m torch.distributed.run ... bench.py --gpus N`, or, with no
launcher, by this script itself: fresh child processes, one per GPU, over RCCL = backend "nccl"): STRONG scaling by
default -- the SAME workload, its genes split into one contiguous range per rank (illico_amd.distributed); the matrix
is generated per 256-gene block from (seed, block), so every N computes the identical problem and no input is ever
exchanged.  The path's one collective -- the gather of every rank's planes to rank 0 -- is INSIDE the timed steps:
step k's gather runs on RCCL's stream under step k + 1's pass, and all K gathers have completed when the clock
stops, so `value` = tests / (pass + gather).  `pass_only` (the same steps without the gather) is reported beside it.
`--scaling weak` gives every rank a full workload-sized gene shard instead.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline       achieved algorithmic GB/s of the dominant kernel (HIP events on the engine's stream, live, inside the
                 timed region) against the 8 TB/s HBM peak; `traffic` = PMC bytes per launch from profiles/traffic.json
  cpu_baseline   the CPU oracle (C restatement of illico's algorithm, oracle/, built -O3 -march=native on this box,
                 threads pinned one per physical core) on a bounded sample of the same workload: all physical cores, and
                 `at_8_threads` (the reference's headline setting, README.md:4)            [N = 1 only]
  parity         16 genes of the LAST timed step's planes compared with the oracle: measured errors, not asserted ones
  timing_scopes  SURVEY.md 8d: (i) engine = ms_per_step, (ii) engine + H2D / D2H transfers, (iii) the drop-in call
  c5_strong      BASELINE configs[4] (1M x 30k x 5k dense OVO) measured in the same launch, genes split over the N
                 ranks: ms per pass, ms per gather, ms per pass + gather (overlapped as above), bytes into rank 0,
                 roofline fraction -- so a driver that only ever runs `bench.py --gpus N` records configs[4] at every N
                 (`--no-c5` skips it; `--c5-cells/--c5-genes/--c5-groups` shrink it for tests)
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
    "c5": dict(cells=1_000_000, genes=30_000, groups=5_000, sparsity=0.5, test="ovo", fmt="dense",
               label="dense {N}x{M}x{G} OVO (BASELINE configs[4], genes sharded over the ranks)"),
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
    ap.add_argument("--genes", type=int, default=None, help="genes in total (--scaling strong) / per GPU (--scaling weak)")
    ap.add_argument("--groups", type=int, default=None)
    ap.add_argument("--sparsity", type=float, default=None)
    ap.add_argument("--test", choices=["ovo", "ovr"], default=None)
    ap.add_argument("--format", choices=["dense", "csc", "csr"], default=None, dest="fmt")
    ap.add_argument("--values", choices=["counts", "continuous", "nb"], default="counts",
                    help="counts: the headline Poisson counts; continuous: log1p(counts * U(0.5,1.5)), the secondary stress of SURVEY.md 8d; "
                         "nb: heavy-tailed counts (log-normal gene means: ~20 %% of the genes beyond 63, ~4 %% beyond 255 -- what a real count "
                         "matrix's highly expressed genes look like)")
    ap.add_argument("--mean-max", type=float, default=15.0, help="gene means ~ U(0.1, mean-max); 15 = the reference's fixture (SURVEY.md 8d)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--gene-batch", type=int, default=0, help="genes per engine pass (0 = auto)")
    ap.add_argument("--gather-batches", type=int, default=1, help="gene blocks per step, each gathered on its own (N>1)")
    ap.add_argument("--no-gather-in-step", action="store_true", help="N>1: time the passes alone; the gather runs once afterwards and is reported apart")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scopes", action="store_true", help="skip timing scopes (ii) and (iii)")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-c5", action="store_true", help="skip the c5_strong object (BASELINE configs[4] in the same launch)")
    ap.add_argument("--no-extras", action="store_true", help="skip the objects c3 / c3_csr / c4 / c2_continuous_ovr (the other single-GPU BASELINE "
                    "configs and one continuous line, measured in the same launch at N = 1)")
    ap.add_argument("--extras-steps", type=int, default=10)
    ap.add_argument("--extras-cpu-seconds", type=float, default=2.0, help="target wall time of each CPU baseline run of the extra configs")
    ap.add_argument("--dtype", choices=["f32", "f64", "f64w"], default="f32",
                    help="value type of the matrix handed to the engine (the workloads are float32: SURVEY.md 8d).  f64: genuine doubles (every non-zero "
                         "value times 1 + 2^-30: no float32 holds it); f64w: the float32 values widened (what the engine may narrow again)")
    ap.add_argument("--c5-cells", type=int, default=1_000_000)
    ap.add_argument("--c5-genes", type=int, default=30_000)
    ap.add_argument("--c5-groups", type=int, default=5_000)
    ap.add_argument("--c5-steps", type=int, default=3)
    ap.add_argument("--no-single-call", action="store_true", help="skip the single_call measurement (one call, gathers of its own blocks only)")
    ap.add_argument("--no-n1-reference", action="store_true", help="N > 1: do not let rank 0 compute the whole workload alone for speedup_vs_n1")
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


GEN_BLOCK = 256  # genes per generator block: block j of a workload is a function of (seed, j) alone


def make_matrix(torch, n_cells, n_genes, sparsity, seed, device, continuous=False, mean_max=15.0, values=None, gene_lb=0):
    """Columns [gene_lb, gene_lb + n_genes) of the workload's matrix: Poisson(gene mean) float32 with `sparsity` of the
    entries zeroed, generated on the device block by block.  A block's generator is seeded by (seed, block number), so a rank
    that owns a gene range computes exactly the columns a single GPU would have."""
    values = values or ("continuous" if continuous else "counts")
    # the row pitch is padded to a multiple of 128 bytes (the C-ABI takes any leading dimension): the fused kernels read
    # 256-byte row segments, which then cover two 128-byte lines instead of straddling three (C5 shard, 3750 genes: a pitch
    # of 15 000 bytes would misalign every row; C2's 32 000 bytes are aligned as they are)
    ld = (n_genes + 31) // 32 * 32
    X = torch.empty((n_cells, ld), dtype=torch.float32, device=device)[:, :n_genes]
    gen = torch.Generator(device=device)
    j = gene_lb
    while j < gene_lb + n_genes:
        blk_no = j // GEN_BLOCK
        b0, b1 = blk_no * GEN_BLOCK, (blk_no + 1) * GEN_BLOCK
        gen.manual_seed(int(seed) * 1_000_003 + blk_no)
        if values == "nb":   # log-normal gene means, median 7.7: P(a gene's largest count > 63) ~ 0.2, P(> 255) ~ 0.04
            m = torch.exp(torch.empty(GEN_BLOCK, device=device).normal_(2.04, 1.8, generator=gen)).clamp_(0.05, 2000.0)
        else:
            m = torch.empty(GEN_BLOCK, device=device).uniform_(0.1, mean_max, generator=gen)
        rates = m.unsqueeze(0).expand(n_cells, GEN_BLOCK).contiguous()
        blk = torch.poisson(rates, generator=gen)
        keep = torch.rand(blk.shape, device=device, generator=gen) >= sparsity
        if values == "continuous":  # normalised-like data: (almost) no ties among the non-zeros
            blk = torch.log1p(blk * torch.empty_like(blk).uniform_(0.5, 1.5, generator=gen))
        blk = blk * keep
        lo, hi = max(j, b0), min(gene_lb + n_genes, b1)
        X[:, lo - gene_lb: hi - gene_lb] = blk[:, lo - b0: hi - b0]
        j = hi
        del rates, blk, keep
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


class Job:
    """One workload on this rank's GPU: data, engine calls, the gather; `measure` times it."""

    def __init__(self, torch, dist, eng, args, *, cells, genes_total, groups, sparsity, test, fmt, values, mean_max, seed,
                 rank, world, device, scaling, dtype="f32"):
        from illico_amd.distributed import rank_gene_range
        self.torch, self.dist, self.eng, self.args = torch, dist, eng, args
        self.rank, self.world, self.device = rank, world, device
        self.N, self.G, self.test, self.fmt = cells, groups, test, fmt
        self.ovr = test == "ovr"
        if scaling == "strong":  # one workload's genes, one contiguous range per rank (equal expected stored entries per gene
            # in this synthetic matrix: ranges by gene count; illico_amd.distributed balances real sparse input by stored entries)
            self.g_lb, self.g_ub = rank_gene_range(genes_total, rank, world)
            self.M_total = genes_total
        else:
            self.g_lb, self.g_ub = rank * genes_total, (rank + 1) * genes_total
            self.M_total = genes_total * world
        self.M = self.g_ub - self.g_lb
        self.codes = make_labels(cells, groups, seed)
        self.grpc = group_container(self.codes, groups, self.ovr)
        self.X = make_matrix(torch, cells, max(self.M, 1), sparsity, seed, device, values=values, mean_max=mean_max, gene_lb=self.g_lb)[:, :self.M]
        self.sparse_fmt = fmt if fmt != "dense" else None
        self.csx, self.nnz, self.X_sample = None, None, None
        if self.sparse_fmt:
            self.X_sample = self.X[:, :min(self.M, 1024)].clone()  # dense copy of the first genes: oracle spot checks / CPU baseline sample
            self.csx = compress(torch, self.X, self.sparse_fmt)
            self.nnz = int(self.csx[0].numel())
            self.X = None
            torch.cuda.empty_cache()
        self.esize = 4
        if dtype in ("f64", "f64w"):  # f64w: the same values, widened (a float32 matrix after a float64 step upstream); f64: no longer float32 values
            self.esize = 8
            scale = 1.0 + 2.0 ** -30 if dtype == "f64" else 1.0
            if self.sparse_fmt:
                self.csx = (self.csx[0].double() * scale, self.csx[1], self.csx[2])
                self.X_sample = self.X_sample.double() * scale
            else:
                self.X = self.X.double() * scale
        torch.cuda.synchronize()
        eng.set_groups(self.grpc)

    def alg_bytes(self):
        """input once + 4 B per cell of codes + three f64 planes (this rank's share)."""
        if self.sparse_fmt:
            return self.nnz * (self.esize + 4) + ((self.M if self.sparse_fmt == "csc" else self.N) + 1) * 4 + 4 * self.N + 24 * self.G * self.M
        return self.N * self.M * self.esize + 4 * self.N + 24 * self.G * self.M

    def setup(self, n_blocks, gather):
        from illico_amd.distributed import rank_gene_range, shard_bounds
        torch = self.torch
        self.gather = gather and self.world > 1
        self.blocks = [b for b in shard_bounds(self.M, max(1, n_blocks)) if b[1] > b[0]] if not self.gather else shard_bounds(self.M, max(1, n_blocks))
        # one contiguous (3, G, w) staging tensor per gene block: the engine writes its planes straight into it.  Two sets, used
        # by alternate steps: a consumer (the gather, or whoever reads the planes) works on step k's set while step k + 1
        # computes into the other -- and with ILLICO_FLAG_DEFER a dense pass is enqueued before the previous one's route flags
        # have been looked at, so there is no host round trip between passes.
        if self.gather:  # every rank's block b must have ONE width (torch.distributed.gather): the widest rank's
            ranges = [rank_gene_range(self.M_total, r, self.world) for r in range(self.world)]
            self.widths = [max(shard_bounds(ub - lb, len(self.blocks))[b][1] - shard_bounds(ub - lb, len(self.blocks))[b][0]
                               for lb, ub in ranges) for b in range(len(self.blocks))]
        else:
            self.widths = [ub - lb for lb, ub in self.blocks]
        self.stage_sets = [[torch.zeros((3, self.G, max(w, 1)), dtype=torch.float64, device=self.device) for w in self.widths] for _ in range(2)]
        self.recv_sets = None
        if self.gather and self.rank == 0:
            self.recv_sets = [[[torch.empty_like(st) for _ in range(self.world)] for st in ss] for ss in self.stage_sets]
        self.step_no = 0
        self.pending = []
        self.stages = self.stage_sets[0]

    def run_block(self, lb, ub, out, defer):
        if self.sparse_fmt:
            self.eng.run_sparse(self.sparse_fmt, self.csx[0], self.csx[1], self.csx[2], (self.N, self.M), lb, ub, out=out, defer=defer)
        else:
            self.eng.run_dense(self.X, lb, ub, out=out, defer=defer)

    def step(self):
        from illico_amd.distributed import gather_block_async
        k = self.step_no & 1
        self.stages = self.stage_sets[k]
        self.step_no += 1
        handles = []
        for b, (lb, ub) in enumerate(self.blocks):
            st = self.stages[b]
            if ub > lb:
                w = ub - lb
                self.run_block(lb, ub, (st[0][:, :w], st[1][:, :w], st[2][:, :w]), defer=not self.gather and not self.args.no_defer)
            if self.gather:  # complete planes (the call above waited for its route flags): hand them to RCCL, do not wait
                handles.append(gather_block_async(st, self.recv_sets[k][b] if self.rank == 0 else None, self.rank, self.world))
        # the gather of step k - 1 ran under this step's pass; its buffers are the next step's
        for h in self.pending:
            h.wait()
        self.pending = handles

    def drain(self):
        for h in self.pending:
            h.wait()
        self.pending = []

    def sync(self):
        self.drain()
        self.eng.synchronize()  # completes a deferred pass (genes the fused route could not take), then waits for the stream
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def timed(self, steps):
        """`steps` steps between two barriers; the MAX over ranks, in seconds."""
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.sync()
        dt = time.perf_counter() - t0
        if self.world > 1:
            red_dev = self.device if self.args.backend == "nccl" else self.torch.device("cpu")
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=red_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def single_call(self, reps=5):
        """ONE call, as a user of asymptotic_wilcoxon makes it (asymptotic_wilcoxon.py:213-241: its gene chunks one after the other):
        this rank's genes in the blocks of `setup`, block b's gather under block b + 1's pass, the clock stopped when rank 0 holds every
        plane (nothing of a NEXT call hides this call's last gather).  Median of `reps` calls, each the MAX over ranks; ms."""
        runs = []
        for _ in range(reps):
            self.sync()
            t0 = time.perf_counter()
            self.step()
            self.sync()
            dt = time.perf_counter() - t0
            if self.world > 1:
                red_dev = self.device if self.args.backend == "nccl" else self.torch.device("cpu")
                t = self.torch.tensor([dt], dtype=self.torch.float64, device=red_dev)
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                dt = float(t.item())
            runs.append(dt * 1e3)
        return float(np.median(runs)), [round(r, 4) for r in runs]

    def host_columns(self, cols):
        """Dense float32 host copy of a few of this rank's genes (whatever the input format)."""
        torch = self.torch
        if not self.sparse_fmt:
            return self.X[:, cols].contiguous().cpu().numpy()
        out = np.zeros((self.N, len(cols)), dtype=np.float32 if self.esize == 4 else np.float64)
        data, indices, indptr = self.csx
        if self.sparse_fmt == "csc":
            ip = indptr.cpu().numpy()
            for i, c in enumerate(cols):
                s, e = int(ip[c]), int(ip[c + 1])
                out[indices[s:e].cpu().numpy(), i] = data[s:e].cpu().numpy()
            return out
        for i, c in enumerate(cols):  # CSR: one pass over the column indices per gene, on the device
            pos = (indices == int(c)).nonzero().flatten()
            rows = torch.searchsorted(indptr[1:].to(torch.int64), pos, right=True)
            out[rows.cpu().numpy(), i] = data[pos].cpu().numpy()
        return out

    def plane_columns(self, cols):
        got = [np.empty((self.G, len(cols))) for _ in range(3)]
        for i, c in enumerate(cols):
            for b, (lb, ub) in enumerate(self.blocks):
                if lb <= c < ub:
                    for k in range(3):
                        got[k][:, i] = self.stages[b][k][:, c - lb].cpu().numpy()
        return got

    def parity(self, n_genes=16):
        import oracle
        cols = sorted(set(np.linspace(0, self.M - 1, min(n_genes, self.M)).astype(int).tolist()))
        want = oracle.run(self.host_columns(cols), self.grpc, batch_size=1, n_threads=min(len(cols), 8))
        got = self.plane_columns(cols)
        mask = np.ones(self.G, dtype=bool)
        if not self.ovr:
            mask[0] = False  # the reference leaves the reference group's row unspecified
        with np.errstate(divide="ignore", invalid="ignore"):
            perr = np.abs(got[0][mask] - want[0][mask]) / np.abs(want[0][mask])
            ferr = np.abs(got[2] - want[2]) / np.abs(want[2])
        perr = np.where((got[0][mask] == want[0][mask]) | (np.isnan(got[0][mask]) & np.isnan(want[0][mask])), 0.0, perr)  # (a group without cells: NaN on both sides)
        ferr = np.where((got[2] == want[2]) | (np.isnan(got[2]) & np.isnan(want[2])), 0.0, ferr)
        return {"genes_checked": cols, "tests_checked": int(mask.sum() * len(cols)),
                "statistic_mismatches": int((got[1][mask] != want[1][mask]).sum()),
                "p_value_max_rel_err": float(np.max(perr)), "fold_change_max_rel_err": float(np.max(ferr)),
                "bar": "statistic exact; p_value and fold_change rtol 1e-12", "against": "oracle/ (CPU restatement pinned to the reference's outputs)"}


def measure(job, steps, warmup, settle=5, events=True):
    """First call, settle + warm-up, one profiled step, the timed steps (gather inside when job.gather); returns a dict."""
    eng = job.eng
    job.sync()
    t0 = time.perf_counter()
    job.step()
    job.drain()
    eng.synchronize()
    first_call_ms = (time.perf_counter() - t0) * 1e3
    # The first ~8 passes of a fresh process run up to 6 % slower than the steady state (tools/ramp.py): a few settling
    # passes before the W warm-up steps, so that a small W still measures the steady state.  Untimed, like the warm-up.
    for _ in range(settle + warmup):
        job.step()
    job.sync()
    # One untimed step with HIP events around every kernel: the per-kernel breakdown, and which kernel dominates.
    # Inside the timed region only that kernel carries events (each event pair drains the stream around a launch).
    eng.profile(True)
    eng.profile_reset()
    job.step()
    job.sync()
    breakdown = eng.profile_get()
    dom = max(breakdown.items(), key=lambda kv: kv[1]["ms"])[0] if breakdown else None
    eng.profile_only(dom)
    eng.profile_reset()
    if not events:  # diagnostic only: the contract wants the dominant kernel timed inside the timed region
        eng.profile(False)
    dt = job.timed(steps)
    prof = eng.profile_get()
    eng.profile(False)
    eng.profile_only(None)
    return {"dt": dt, "ms_per_step": dt / steps * 1e3, "first_call_ms": first_call_ms, "breakdown": breakdown, "dom": dom, "prof": prof,
            "settle": settle}


def single_call_of(torch, dist, eng, args, job, job_kw, n_blocks, tests, steady_ms, gather_alone_ms=None):
    """The `single_call` object of a workload (see Job.single_call) and, at N > 1, its N = 1 reference: the SAME workload computed by
    rank 0 alone in the same launch (the other ranks wait), so that the line carries its own speed-up.  Collective: every rank calls."""
    world, rank = job.world, job.rank
    job.sync()
    job.setup(n_blocks if world > 1 else 1, gather=world > 1)  # (one GPU: nothing to overlap, the call is one pass over every column)
    for _ in range(2):
        job.step()
    ms, runs = job.single_call()
    out = {"blocks": len(job.blocks), "ms_single_call": round(ms, 4), "runs_ms": runs, "tests_per_s": round(tests / (ms * 1e-3), 1),
           "steady_state_ms_per_step": round(steady_ms, 4),
           "note": "one call: the rank's genes in `blocks` blocks, block b's gather under block b + 1's pass, clock stopped when rank 0 holds every "
                   "plane (max over ranks, median of the runs); steady_state = steps back to back, step k's gather under step k + 1's pass"}
    if world > 1:
        bytes0 = 24 * job.G * (job.M_total - job.M)
        if gather_alone_ms is None:  # one gather by itself (no pass under it)
            from illico_amd.distributed import gather_block_async
            job.sync()
            t0 = time.perf_counter()
            hs = [gather_block_async(st, job.recv_sets[0][b] if rank == 0 else None, rank, world) for b, st in enumerate(job.stage_sets[0])]
            for h in hs:
                h.wait()
            torch.cuda.synchronize(); dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=job.device if args.backend == "nccl" else torch.device("cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gather_alone_ms = float(t.item()) * 1e3
            out["ms_gather_alone"] = round(gather_alone_ms, 4)
        if gather_alone_ms:
            out["gather_into_rank0_GBs"] = round(bytes0 / (gather_alone_ms * 1e-3) / 1e9, 1)
            out["gather_ceiling_note"] = ("24 bytes per test travel into rank 0: at the measured gather rate the planes alone take "
                                          f"{bytes0 / 1e6:.0f} MB / rate; a call cannot be faster than its last block's gather")
        n1 = None
        if not args.no_n1_reference and args.scaling == "strong":
            if rank == 0:
                try:
                    j1 = Job(torch, dist, eng, args, rank=0, world=1, scaling="strong", **job_kw)
                    j1.setup(1, gather=False)
                    for _ in range(3):
                        j1.step()
                    n1_ms, n1_runs = j1.single_call()
                    n1 = {"ms_single_call": round(n1_ms, 4), "runs_ms": n1_runs, "genes": j1.M,
                          "note": "the same workload on rank 0 alone in the same launch: one pass over every column (what a single GPU's call is)"}
                    del j1
                    torch.cuda.empty_cache()
                    eng.set_groups(job.grpc)
                except (MemoryError, RuntimeError) as e:
                    n1 = {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}
            dist.barrier()
        out["n1_reference"] = n1
        if n1 and "ms_single_call" in n1:
            out["speedup_vs_n1"] = round(n1["ms_single_call"] / ms, 3)
    # ---- the drop-in scope of the same call: the planes on the HOST.  Every rank's engine writes host planes that are its own column
    # range of ONE shared [3][G][M] result (illico_amd.distributed.SharedHostPlanes: what asymptotic_wilcoxon_sharded's default tail
    # does) -- N PCIe links side by side, nothing over xGMI -- the clock stopped when every rank's planes have landed.
    try:
        from illico_amd.distributed import SharedHostPlanes, shard_bounds
        shared = SharedHostPlanes(job.G, job.M_total, group=None)
        nb = max(1, n_blocks) if world > 1 else 1   # (one GPU: the call is one pass over every column, as above)

        def to_host_once():
            job.sync()
            t0 = time.perf_counter()
            for lb, ub in shard_bounds(job.M, nb):
                if ub > lb:
                    job.run_block(lb, ub, shared.columns(job.g_lb + lb, job.g_lb + ub), defer=False)
            job.eng.synchronize()
            shared.barrier()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=job.device if args.backend == "nccl" else torch.device("cpu"))
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt * 1e3

        first = to_host_once()          # (the mapping's pages are touched here)
        runs = [to_host_once() for _ in range(3)]
        med = float(np.median(runs))
        out["ms_to_host"] = round(med, 3)
        out["to_host"] = {"runs_ms": [round(r, 3) for r in runs], "first_ms_fresh_mapping": round(first, 3), "blocks_per_rank": nb,
                          "bytes_to_host_per_rank": 24 * job.G * job.M, "bytes_to_host_total": 24 * job.G * job.M_total,
                          "aggregate_GBs": round(24 * job.G * job.M_total / (med * 1e-3) / 1e9, 1),
                          "note": "one call, planes on the host: each rank's engine writes its own column range of one shared host result over its "
                                  "own PCIe link (no gather); max over ranks, median of 3"}
        del shared
    except (OSError, MemoryError, RuntimeError) as e:
        out["to_host"] = {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}
    return out


def roofline_of(job, m, steps, wl_key):
    dom, prof = m["dom"], m["prof"]
    if not dom or dom not in prof:
        return None
    alg = job.alg_bytes()
    launches = prof[dom]["launches"]
    avg_ms = prof[dom]["ms"] / launches
    bytes_per_launch = alg * steps / launches
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
    traffic = None
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists() and wl_key is not None:
        try:  # PMC traffic is only valid for the workload it was collected on
            for ent in json.loads(tf.read_text()).get("entries", []):
                if ent.get("kernel_id") == dom and all(ent.get("workload", {}).get(k) == v for k, v in wl_key.items()) \
                        and launches == steps * ent.get("launches_per_step", 1):
                    traffic = ent.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    pipe = alg / (m["pass_ms"] * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches / steps,
            "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "all_kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in m["breakdown"].items()},
            "all_kernels_note": "one untimed step with events around every kernel, taken before the timed region",
            "pipeline_achieved": round(pipe, 2), "pipeline_frac": round(pipe / HBM_PEAK_GBS, 5),
            "pipeline_note": "this rank's algorithmic bytes over its pass time (the gather excluded)"}


def cpu_baseline_of(job, seconds):
    """The oracle (C restatement of illico's algorithm, oracle/, -O3 -march=native, OpenMP over gene chunks, threads pinned one per
    physical core) on a bounded sample of `job`'s workload -- all physical cores, and the reference's headline 8 threads -- each run
    sized to ~`seconds` of wall time.  Test infrastructure timed as a reported baseline: nothing of it is in the product path."""
    import oracle
    from scipy import sparse as sp
    oracle.use_native(True)
    n_phys = oracle.pin_threads(True)
    N, M, G, sparse_fmt = job.N, job.M, job.G, job.sparse_fmt
    M_s = M if not sparse_fmt else int(job.X_sample.shape[1])

    def sample(ns):
        if sparse_fmt:  # the same stored entries in the same format, from the dense copy of the first genes
            Xs = job.X_sample[:, :ns].contiguous().cpu().numpy()
            return sp.csc_matrix(Xs) if sparse_fmt == "csc" else sp.csr_matrix(Xs)
        return job.X[:, :ns].contiguous().cpu().numpy()

    def timed(n_threads, batch):
        """Probe with one batch per thread, then a run sized to ~`seconds`; returns (tests/s, genes, wall)."""
        ns = min(M_s, n_threads * batch)
        Xs = sample(ns)
        t1 = time.perf_counter()
        oracle.run(Xs, job.grpc, batch_size=batch, n_threads=n_threads)
        el = time.perf_counter() - t1
        ns2 = int(min(M_s, (G * ns / el) * seconds / G))
        ns2 = max(n_threads * batch, (ns2 // (n_threads * batch)) * (n_threads * batch))
        if ns2 > ns:
            ns = min(ns2, M_s)
            Xs = sample(ns)
            t1 = time.perf_counter()
            oracle.run(Xs, job.grpc, batch_size=batch, n_threads=n_threads)
            el = time.perf_counter() - t1
        return G * ns / el, ns, el

    # all physical cores: chunks wide enough that the row gathers use whole cache lines, narrow enough to keep every
    # core busy within the sample; 8 threads: the reference's 256-gene chunks (README.md:124 benchmarks)
    b_all = int(max(8, min(256, M_s // max(n_phys, 1))))
    v_all, ns_all, el_all = timed(n_phys, b_all)
    v_8, ns_8, el_8 = timed(min(8, n_phys), int(min(256, max(8, M_s // 8))))
    oracle.pin_threads(False)
    oracle.use_native(False)
    what = (f"same {N}x{M}x{G} {job.test.upper()} {job.fmt} workload; oracle/ (C restatement of illico's algorithm), -O3 -march=native, "
            "OpenMP over gene chunks, threads pinned one per physical core")
    return {"value": round(v_all, 1), "unit": "tests/s", "cores": n_phys, "kind": "port",
            "sample": f"first {ns_all} genes in chunks of {b_all}, {el_all:.1f}s wall; {what}",
            "at_8_threads": {"value": round(v_8, 1), "cores": min(8, n_phys), "sample": f"first {ns_8} genes, {el_8:.1f}s wall",
                             "note": "the reference's headline setting (README.md:4: 8 threads)"},
            "logical_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}


# the other single-GPU BASELINE configs and one continuous line, at the headline's shape, measured in the SAME launch (N = 1): the
# reference's own benchmark matrix is dense + CSR x OVO + OVR (tests/test_asymptotic_wilcoxon.py:276-340)
EXTRAS = {
    "c3": dict(fmt="csc", test="ovo", sparsity=0.9, values="counts", workload="c3", label="BASELINE configs[2]: the headline's shape as CSC, 90 % zeros, OVO"),
    "c3_csr": dict(fmt="csr", test="ovo", sparsity=0.9, values="counts", workload="c3", label="configs[2]'s matrix as CSR (AnnData's default container), OVO"),
    "c4": dict(fmt="dense", test="ovr", sparsity=0.5, values="counts", workload="c4", label="BASELINE configs[3]: the headline's matrix, one-versus-rest"),
    "c2_continuous_ovr": dict(fmt="dense", test="ovr", sparsity=0.5, values="continuous", workload="c2",
                              label="the headline's shape with log-normalised values (SURVEY.md 8d's secondary stress), one-versus-rest"),
}


def extra_config(torch, dist, eng, args, device, tag):
    """One more workload of the headline's shape in this launch: ms_per_step, roofline (dominant kernel timed live), parity, cpu_baseline."""
    e = EXTRAS[tag]
    t0 = time.perf_counter()
    job = Job(torch, dist, eng, args, cells=args.cells, genes_total=args.genes, groups=args.groups, sparsity=e["sparsity"], test=e["test"], fmt=e["fmt"],
              values=e["values"], mean_max=15.0, seed=args.seed, rank=0, world=1, device=device, scaling="strong")
    t_gen = time.perf_counter() - t0
    job.setup(1, gather=False)
    m = measure(job, args.extras_steps, 1, settle=3)
    m["pass_ms"] = m["ms_per_step"]
    wl_key = {"workload": e["workload"], "cells": job.N, "genes_per_gpu": job.M, "groups": job.G, "test": e["test"], "format": e["fmt"],
              "values": e["values"], "sparsity": e["sparsity"]}
    r = roofline_of(job, m, args.extras_steps, wl_key)
    keep = ("kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "algorithmic_bytes_per_launch",
            "pipeline_frac", "all_kernels_ms_per_step")
    out = {"workload": f"{e['fmt']} {job.N}x{job.M}x{job.G} {e['test'].upper()}" + ("" if e["values"] == "counts" else f" [{e['values']} values]"),
           "what": e["label"], "format": e["fmt"], "test": e["test"], "values": e["values"], "sparsity": e["sparsity"], "nnz": job.nnz,
           "steps": args.extras_steps, "ms_per_step": round(m["ms_per_step"], 4),
           "tests_per_s": round(job.G * job.M / (m["ms_per_step"] * 1e-3), 1),
           "roofline": None if r is None else dict({"bound": "hbm"}, **{k: r[k] for k in keep}),
           "parity": None if args.no_parity else job.parity(16),
           "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline_of(job, args.extras_cpu_seconds),
           "generate_s": round(t_gen, 1)}
    del job
    torch.cuda.empty_cache()
    return out


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

    eng = Engine(local_rank)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.gene_batch:
        eng.set_option("gene_batch", args.gene_batch)
    for kv in args.engine_option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))

    N, G = args.cells, args.groups
    job_kw = dict(cells=N, genes_total=args.genes, groups=G, sparsity=args.sparsity, test=args.test, fmt=args.fmt, values=args.values,
                  mean_max=args.mean_max, seed=args.seed, device=device, dtype=args.dtype)
    job = Job(torch, dist, eng, args, rank=rank, world=world, scaling=args.scaling, **job_kw)
    M, M_total, ovr, sparse_fmt, nnz = job.M, job.M_total, job.ovr, job.sparse_fmt, job.nnz
    in_step = world > 1 and not args.no_gather_in_step
    job.setup(args.gather_batches, gather=in_step)

    m = measure(job, args.steps, args.warmup, events=not args.no_events)
    ms_per_step = m["ms_per_step"]
    tests_per_step = G * M_total
    value = tests_per_step / (m["dt"] / args.steps)

    # N > 1: the same steps without the gather (pass only), and -- when the gather is not in the step -- the gather alone
    pass_only_ms, gather_ms = ms_per_step, None
    if world > 1:
        if in_step:
            job.sync()
            job.setup(args.gather_batches, gather=False)
            for _ in range(2):
                job.step()
            pass_only_ms = job.timed(args.steps) / args.steps * 1e3
            job.setup(args.gather_batches, gather=True)  # (the planes of the last gathered step are recomputed for the parity leg)
            job.step()
            job.sync()
        else:
            job.setup(args.gather_batches, gather=True)
            job.step(); job.sync()           # RCCL connections set up
            t0 = time.perf_counter()
            job.step(); job.sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if args.backend == "nccl" else torch.device("cpu"))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gather_ms = max(0.0, float(t.item()) * 1e3 - ms_per_step)
    m["pass_ms"] = pass_only_ms
    single = None
    if not args.no_single_call:
        single = single_call_of(torch, dist, eng, args, job, job_kw, max(args.gather_batches, 4), tests_per_step, ms_per_step)

    # what this box's HBM delivers to a plain streaming read (boxes of the pool differ by up to 15 %: 1.65 vs 1.90 ms for the
    # same C2 kernel): torch's sum over the resident input
    calib = None
    if rank == 0 and M > 0:
        src = job.X if job.X is not None else job.csx[0]
        src.sum(); torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(5):
            src.sum()
        torch.cuda.synchronize()
        calib = {"stream_read_GBs": round(src.numel() * src.element_size() * 5 / (time.perf_counter() - tc) / 1e9, 1),
                 "what": "torch .sum() over the resident input, 5 passes: the box's own streaming-read rate, for comparing runs on different boxes"}

    result = None
    if rank == 0:
        wl_key = {"workload": args.workload, "cells": N, "genes_per_gpu": M, "groups": G, "test": args.test, "format": args.fmt,
                  "values": args.values, "sparsity": args.sparsity}
        if args.mean_max != 15.0:
            wl_key["mean_max"] = args.mean_max
        if args.dtype != "f32":
            wl_key["dtype"] = args.dtype
        roofline = roofline_of(job, m, args.steps, wl_key)

        # ---- parity of what was just timed: 16 genes of the final step's planes against the oracle ----
        parity = None if args.no_parity or M == 0 else job.parity(16)

        # ---- CPU baseline: the oracle on this box's host cores, bounded sample, threads pinned ----
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # contract: rank 0 at N = 1 only
            cpu = cpu_baseline_of(job, args.cpu_seconds)

        # ---- timing scopes (ii) engine + transfers and (iii) the drop-in call, SURVEY.md 8d ----
        scopes = {"engine_ms": round(ms_per_step, 4), "first_call_ms": round(m["first_call_ms"], 3),
                  "claim": "the >= 100x-over-CPU target of BASELINE.json is an engine-scope (input resident in HBM) figure; (ii) and (iii) are PCIe-bound"}
        if not args.no_scopes and world == 1:
            try:
                import pandas as pd
                from illico_amd import AnnDataLite, asymptotic_wilcoxon
                csx = job.csx
                if sparse_fmt:
                    from scipy import sparse as sp
                    ctor = sp.csc_matrix if sparse_fmt == "csc" else sp.csr_matrix
                    Xh = ctor((csx[0].cpu().numpy(), csx[1].cpu().numpy(), csx[2].cpu().numpy()), shape=(N, M))
                    run_host = lambda: eng.run_sparse(sparse_fmt, Xh.data, Xh.indices, Xh.indptr, (N, M), 0, M)
                else:
                    Xh = job.X.contiguous().cpu().numpy()
                    run_host = lambda: eng.run_dense(Xh, 0, M)
                run_host()  # scratch for the staged route
                runs, held = [], []  # the planes stay alive until the clock has stopped, as a caller's would: returning 384 MB
                for _ in range(3):   # to the OS (munmap with the engine's host threads alive) costs 20 - 50 ms of its own
                    t1 = time.perf_counter()
                    held.append(run_host())
                    runs.append(round((time.perf_counter() - t1) * 1e3, 2))
                del held
                scopes["engine_plus_transfers_ms"] = sorted(runs)[1]
                scopes["engine_plus_transfers_runs_ms"] = runs
                codes = job.codes
                labels = np.where(codes == 0, "non-targeting", np.char.add("pert_", np.char.zfill(codes.astype(str), 5)))
                adata = AnnDataLite(Xh, obs=pd.DataFrame({"pert": labels}))
                kw = dict(is_log1p=False, group_keys="pert", reference=None if ovr else "non-targeting")
                runs, held = [], []
                for _ in range(3):
                    t1 = time.perf_counter()
                    df = asymptotic_wilcoxon(adata, **kw)
                    runs.append(round((time.perf_counter() - t1) * 1e3, 2))
                    held.append(df)
                del held
                scopes["drop_in_call_ms"] = sorted(runs)[1]
                scopes["drop_in_call_runs_ms"] = runs
                scopes["drop_in_rows"] = int(len(df))
                scopes["note"] = ("(ii) host-resident input -> host planes: H2D of the matrix and D2H of 24 B per test included; "
                                  "(iii) illico_amd.asymptotic_wilcoxon(adata, ...) on the same host matrix: group encoding, (ii), DataFrame assembly; "
                                  "medians of three runs")
                del df, adata, Xh
            except MemoryError as e:  # a host too small for a second copy of the workload
                scopes["skipped"] = f"host memory: {e}"

        result = {
            "metric": "(group x gene) tests/sec", "value": round(value, 1), "unit": "tests/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload]["label"].format(N=N, M=M_total, G=G) + ("" if args.values == "counts" else f" [{args.values} values]"),
                       "workload_id": args.workload, "cells": N, "genes_per_gpu": M, "genes_total": M_total, "groups": G, "format": args.fmt,
                       "test": args.test, "sparsity": args.sparsity, "values": args.values, "gene_mean_max": args.mean_max, "nnz_per_gpu": nnz,
                       "settle_steps": m["settle"],
                       "parallelism": f"gene-shard x{world}, planes gathered to rank 0" if world > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "timing_scopes": scopes, "box_calibration": calib,
            "single_call": single,
            "steady_state": {"ms_per_step": round(ms_per_step, 4), "tests_per_s": round(value, 1),
                             "note": "`value`: the timed steps back to back" + (", step k's gather under step k + 1's pass" if in_step else "")},
        }
        if world > 1:
            plane_bytes = 24 * G * M
            result["pass_only"] = {"ms_per_step": round(pass_only_ms, 4), "tests_per_s": round(tests_per_step / (pass_only_ms * 1e-3), 1),
                                   "note": "the same steps without the gather: every rank's planes stay in its own HBM"}
            if in_step:
                result["final_gather"] = {"in_timed_step": True, "blocks_per_step": len(job.blocks), "bytes_per_rank_per_step": plane_bytes,
                                          "bytes_into_rank0_per_step": 24 * G * (M_total - M),
                                          "note": "value = tests / (pass + gather): step k's gather (torch.distributed.gather over RCCL) runs under step "
                                                  "k + 1's pass, every gather has completed when the clock stops"}
            else:
                result["final_gather"] = {"in_timed_step": False, "ms": round(gather_ms, 3), "bytes_per_rank": plane_bytes,
                                          "bytes_into_rank0": 24 * G * (M_total - M),
                                          "one_pass_plus_gather_ms": round(ms_per_step + gather_ms, 3),
                                          "tests_per_s_pass_plus_gather": round(tests_per_step / ((ms_per_step + gather_ms) * 1e-3), 1)}

    del job
    torch.cuda.empty_cache()

    # ---- the other single-GPU BASELINE configs (and one continuous line) in the same launch ----
    if world == 1 and not args.no_extras:
        for tag in EXTRAS:
            try:
                result[tag] = extra_config(torch, dist, eng, args, device, tag)
            except (MemoryError, RuntimeError) as e:
                result[tag] = {"skipped": f"{type(e).__name__}: {str(e)[:300]}"}

    # ---- BASELINE configs[4] in the same launch: 1M x 30k x 5k dense OVO, genes split over the ranks ----
    if not args.no_c5:
        c5 = None
        try:
            t_gen = time.perf_counter()
            j5 = Job(torch, dist, eng, args, cells=args.c5_cells, genes_total=args.c5_genes, groups=args.c5_groups, sparsity=0.5, test="ovo",
                     fmt="dense", values="counts", mean_max=15.0, seed=args.seed + 5, rank=rank, world=world, device=device, scaling="strong")
            t_gen = time.perf_counter() - t_gen
            j5.setup(1, gather=False)
            m5 = measure(j5, args.c5_steps, 1, settle=2)
            m5["pass_ms"] = m5["ms_per_step"]
            both_ms, gather_alone_ms = m5["ms_per_step"], 0.0
            if world > 1:
                j5.sync()
                j5.setup(1, gather=True)
                for _ in range(2):
                    j5.step()
                both_ms = j5.timed(args.c5_steps) / args.c5_steps * 1e3
                j5.sync()
                t0 = time.perf_counter()     # one gather by itself (no pass under it)
                from illico_amd.distributed import gather_block_async
                hs = [gather_block_async(st, j5.recv_sets[0][b] if rank == 0 else None, rank, world) for b, st in enumerate(j5.stage_sets[0])]
                for h in hs:
                    h.wait()
                torch.cuda.synchronize(); dist.barrier()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if args.backend == "nccl" else torch.device("cpu"))
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                gather_alone_ms = float(t.item()) * 1e3
            single5 = None
            if not args.no_single_call:
                c5_kw = dict(cells=args.c5_cells, genes_total=args.c5_genes, groups=args.c5_groups, sparsity=0.5, test="ovo", fmt="dense", values="counts",
                             mean_max=15.0, seed=args.seed + 5, device=device)
                single5 = single_call_of(torch, dist, eng, args, j5, c5_kw, max(args.gather_batches, 4), args.c5_groups * args.c5_genes, both_ms,
                                         gather_alone_ms if world > 1 else None)
            if rank == 0:
                tests5 = args.c5_groups * args.c5_genes
                r5 = roofline_of(j5, m5, args.c5_steps, None)
                c5 = {"workload": WORKLOADS["c5"]["label"].format(N=args.c5_cells, M=args.c5_genes, G=args.c5_groups), "scaling": "strong",
                      "cells": args.c5_cells, "genes_total": args.c5_genes, "genes_per_gpu": j5.M, "groups": args.c5_groups, "steps": args.c5_steps,
                      "ms_pass": round(m5["ms_per_step"], 4), "ms_gather_alone": round(gather_alone_ms, 4), "ms_pass_plus_gather": round(both_ms, 4),
                      "tests_per_s": round(tests5 / (both_ms * 1e-3), 1), "tests_per_s_pass_only": round(tests5 / (m5["ms_per_step"] * 1e-3), 1),
                      "bytes_into_rank0": 24 * args.c5_groups * (args.c5_genes - j5.M), "input_bytes_per_gpu": int(args.c5_cells) * j5.M * 4,
                      "roofline": None if r5 is None else {k: r5[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "pipeline_frac", "all_kernels_ms_per_step")},
                      "parity": None if args.no_parity else j5.parity(8), "generate_s": round(t_gen, 1), "single_call": single5,
                      "note": "tests_per_s = tests / (pass + gather), the gather of step k under the pass of step k + 1; at N = 1 there is no gather"}
        except (MemoryError, RuntimeError) as e:
            if rank == 0:
                c5 = {"skipped": f"{type(e).__name__}: {str(e)[:300]}"}
        if rank == 0:
            result["c5_strong"] = c5

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
