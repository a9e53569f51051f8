#!/usr/bin/env python
"""How much of k_ovo_fused's time is the row gather?  Same C2 data and group sizes, cells either shuffled over the
groups (the bench's layout) or sorted by group (indices = identity: every group is a block of consecutive rows)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, 0.5, 0, dev)
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
codes = make_labels(N, G, 0)
for name, cd in (("shuffled", codes), ("sorted", np.sort(codes)), ("shuffled", codes), ("sorted", np.sort(codes))):
    eng.set_groups(group_container(cd, G, False))
    eng.run_dense(X, 0, M, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(5): eng.run_dense(X, 0, M, out=out)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"{name}: median {np.median(ts):.4f} ms per step")
