#!/usr/bin/env python
"""Per-step time of the first steps of a fresh process (clock ramp / first-touch effects)."""
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
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
torch.cuda.synchronize(); time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)
ts = []
for i in range(40):
    t0 = time.perf_counter(); eng.run_dense(X, 0, M, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("steps 0-9  :", [round(t, 3) for t in ts[:10]])
print("steps 10-19:", [round(t, 3) for t in ts[10:20]])
print("steps 30-39:", [round(t, 3) for t in ts[30:]])
