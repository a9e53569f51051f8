#!/usr/bin/env python
"""The C2 step has a fast and a slow mode (about 4 % apart) that depends on where the engine's small arrays and the
output planes land in HBM.  One process, the same X: engine + planes re-created several times with pads of different
sizes in between; each instance is timed twice.  (On the run recorded in DESIGN.md instances 2, 5 and 6 of 10 were
slow, reproducibly; a dynamic per-tile group queue, which removes the repeated reads of the reference tables, did
not change that.)"""
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
grp = group_container(make_labels(N, G, 0), G, False)
def t(eng, out):
    eng.run_dense(X, 0, M, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): eng.run_dense(X, 0, M, out=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3
keep = []
for i in range(10):
    eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream); eng.set_groups(grp)
    out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
    t(eng, out)
    print(f"instance {i}: {t(eng, out):.4f} {t(eng, out):.4f} ms   planes at {[hex(x.data_ptr() >> 20) for x in out]} (MiB)")
    keep.append((eng, out, torch.empty(int(np.random.RandomState(i).randint(1, 64)) * 1024 * 1024, dtype=torch.uint8, device=dev)))
