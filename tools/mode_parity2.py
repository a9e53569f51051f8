#!/usr/bin/env python
"""Follow-up of tools/mode_parity.py: the duration of k_ovo_fused follows WHICH allocation holds the output planes.  Does the
moment of allocation matter?  Plane sets allocated before anything else in the process, after X, after some allocator churn, and
carved out of one early pool.  Usage: python tools/mode_parity2.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
S = 3 * G * M * 8
early = [torch.empty((3, G, M), dtype=torch.float64, device=dev) for _ in range(2)]
pool = torch.empty(2 * S, dtype=torch.uint8, device=dev)
pooled = [pool[i * S:(i + 1) * S].view(torch.float64).view(3, G, M) for i in range(2)]
X = make_matrix(torch, N, M, 0.5, 0, dev)
after_x = [torch.empty((3, G, M), dtype=torch.float64, device=dev) for _ in range(2)]
junk = [torch.empty(int(1.3e9), dtype=torch.uint8, device=dev) for _ in range(6)]
del junk[::2]
torch.cuda.empty_cache()
churn = [torch.empty((3, G, M), dtype=torch.float64, device=dev) for _ in range(2)]
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
def one(st):
    eng.profile(True); eng.profile_reset()
    eng.run_dense(X, 0, M, out=(st[0], st[1], st[2]))
    eng.synchronize(); torch.cuda.synchronize()
    p = eng.profile_get(); eng.profile(False)
    return p["k_ovo_fused"]["ms"] / p["k_ovo_fused"]["launches"]
for _ in range(8): one(early[0])
for rep in range(2):
    for name, group in (("allocated first", early), ("one early pool", pooled), ("after X", after_x), ("after churn", churn)):
        for i, st in enumerate(group):
            ts = [one(st) for _ in range(6)]
            print(f"{name:<16} {i} at {hex(st.data_ptr())}: " + " ".join(f"{t:.3f}" for t in ts))
