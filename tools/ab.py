#!/usr/bin/env python
"""Interleaved A/B of one engine option on the C2 workload in ONE process (same device, same data):
   python tools/ab.py <option> <v1> <v2> ... [--test ovr] [--rounds 6]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
ap = argparse.ArgumentParser(); ap.add_argument("option"); ap.add_argument("values", nargs="+", type=int)
ap.add_argument("--test", default="ovo"); ap.add_argument("--rounds", type=int, default=6); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, 0.5, 0, dev)
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, a.test == "ovr"))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
# calibration of the box: streaming read of the same matrix through torch (boxes differ by several per cent)
torch.cuda.synchronize(); X.sum(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): X.sum()
torch.cuda.synchronize()
print(f"calibration: torch X.sum() reads {X.numel() * 4 / ((time.perf_counter() - t0) / 5) / 1e12:.2f} TB/s")
res = {v: [] for v in a.values}
for r in range(a.rounds + 1):
    for v in a.values:
        eng.set_option(a.option, v)
        eng.run_dense(X, 0, M, out=out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps): eng.run_dense(X, 0, M, out=out)
        torch.cuda.synchronize()
        if r: res[v].append((time.perf_counter() - t0) / a.steps * 1e3)
for v, ts in res.items():
    print(f"{a.option}={v}: median {np.median(ts):.4f} ms  min {min(ts):.4f}  all {[round(t, 3) for t in ts]}")
