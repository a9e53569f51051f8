#!/usr/bin/env python
"""Does the step time depend on WHERE the input sits in HBM?  Same process, same data, three copies of X allocated at
different times (fresh hipMalloc each: torch's cache is emptied in between), timed alternately."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X0 = make_matrix(torch, N, M, 0.5, 0, dev)
pad = torch.empty(3 * 1024**3 + 12345, dtype=torch.uint8, device=dev)   # shifts the next allocation
X1 = X0.clone()
del pad; torch.cuda.empty_cache()
X2 = X0.clone()
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
outs = [tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3)) for _ in range(2)]
res = {}
for r in range(4):
    for xi, X in enumerate((X0, X1, X2)):
        for oi, out in enumerate(outs):
            eng.run_dense(X, 0, M, out=out); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): eng.run_dense(X, 0, M, out=out)
            torch.cuda.synchronize()
            res.setdefault((xi, oi), []).append((time.perf_counter() - t0) / 5 * 1e3)
for k, v in sorted(res.items()):
    print(f"X{k[0]} (ptr %x) out{k[1]}: median {np.median(v):.4f} ms" % (X0, X1, X2)[k[0]].data_ptr())
