#!/usr/bin/env python
"""Interleaved A/B of one engine option on the C3 workload (CSC / CSR, 90 % zeros) in ONE process, with the per-kernel times of each value:
   python tools/ab_sparse.py <option> <v1> <v2> ... [--format csr] [--test ovr] [--rounds 4] [--sparsity 0.9] [--groups 2000]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import compress, group_container, make_labels, make_matrix
from illico_amd._lib import Engine
ap = argparse.ArgumentParser(); ap.add_argument("option"); ap.add_argument("values", nargs="+", type=int)
ap.add_argument("--format", default="csr"); ap.add_argument("--test", default="ovo"); ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--sparsity", type=float, default=0.9); ap.add_argument("--groups", type=int, default=2000)
ap.add_argument("--genes", type=int, default=8000); ap.add_argument("--cells", type=int, default=300_000)
a = ap.parse_args()
N, M, G = a.cells, a.genes, a.groups
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, a.sparsity, 0, dev)
csx = compress(torch, X, a.format)
del X; torch.cuda.empty_cache()
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, a.test == "ovr"))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
def step(): eng.run_sparse(a.format, csx[0], csx[1], csx[2], (N, M), 0, M, out=out, defer=True)
res = {v: [] for v in a.values}; prof = {}
for r in range(a.rounds + 1):
    for v in a.values:
        eng.set_option(a.option, v)
        step(); eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps): step()
        eng.synchronize(); torch.cuda.synchronize()
        if r: res[v].append((time.perf_counter() - t0) / a.steps * 1e3)
        if r == a.rounds:
            eng.set_option("profile", 1); eng.profile_reset(); step(); eng.synchronize(); torch.cuda.synchronize()
            prof[v] = {k: round(x["ms"], 4) for k, x in eng.profile_get().items()}; eng.set_option("profile", 0)
for v, ts in res.items():
    print(f"{a.option}={v}: median {np.median(ts):.4f} ms  min {min(ts):.4f}  kernels {prof[v]}")
