#!/usr/bin/env python
"""Is the allocation that makes k_ovo_fused slow (tools/mode_parity.py) slow for translation-bound accesses in general?  Per plane
set: the fused pass, a linear fill, and random 8-byte scatters / gathers over the set (4M indices: bound by address translation
when the mapping uses small fragments).  Usage: python tools/mode_pages.py"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, 0.5, 0, dev)
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
sets = [torch.empty((3, G, M), dtype=torch.float64, device=dev) for _ in range(6)]
gen = torch.Generator(device=dev); gen.manual_seed(1)
idx = torch.randint(0, 3 * G * M, (1 << 22,), device=dev, generator=gen)
val = torch.ones(1 << 22, dtype=torch.float64, device=dev)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
def fused(st):
    eng.profile(True); eng.profile_reset()
    for _ in range(4): eng.run_dense(X, 0, M, out=(st[0], st[1], st[2]))
    eng.synchronize(); torch.cuda.synchronize()
    p = eng.profile_get(); eng.profile(False)
    return p["k_ovo_fused"]["ms"] / p["k_ovo_fused"]["launches"]
for _ in range(6): fused(sets[0])
for rep in range(2):
    for i, st in enumerate(sets):
        flat = st.view(-1)
        print(f"set {i} at {hex(st.data_ptr())}: fused {fused(st):.3f} ms   fill {timed(lambda: st.zero_()):.3f} ms   "
              f"scatter {timed(lambda: flat.index_put_((idx,), val)):.3f} ms   gather {timed(lambda: flat.index_select(0, idx)):.3f} ms")
