#!/usr/bin/env python
"""One process = one sample of the C2 step's mode: prints k_ovo_fused's HIP-event time and where the buffers landed."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
pre = int(os.environ.get("PRE_MB", "0"))
pad = torch.empty(pre << 20, dtype=torch.uint8, device=dev) if pre else None
X = make_matrix(torch, N, M, 0.5, 0, dev)
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
for _ in range(8): eng.run_dense(X, 0, M, out=out)
torch.cuda.synchronize()
eng.profile(True); eng.profile_reset()
for _ in range(10): eng.run_dense(X, 0, M, out=out)
torch.cuda.synchronize()
p = eng.profile_get()
free, total = torch.cuda.mem_get_info()
print(f"k_ovo_fused {p['k_ovo_fused']['ms'] / p['k_ovo_fused']['launches']:.4f} ms  X at {hex(X.data_ptr())} (mod 1GiB {X.data_ptr() % (1 << 30) >> 20} MiB)  out {hex(out[0].data_ptr())}  free {free >> 20} MiB", flush=True)
