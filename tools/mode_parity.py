#!/usr/bin/env python
"""The C2 step alternates between two durations when bench.py alternates its two sets of output planes (rocprofv3 kernel trace:
1.655 / 1.860 ms by step parity).  Which of the things that alternate is it?  One process, one X; k_ovo_fused timed per call
(HIP events) for: each plane set on its own, the sets alternating, deferred and not, planes carved from one pool, and a third and
fourth set.  Usage: python tools/mode_parity.py"""
import sys
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
sets = [torch.empty((3, G, M), dtype=torch.float64, device=dev) for _ in range(4)]
for i, s in enumerate(sets):
    print(f"set {i} at {hex(s.data_ptr())}  (mod 2 MiB: {s.data_ptr() % (2 << 20)}, mod 1 GiB: {hex(s.data_ptr() % (1 << 30))})")
print(f"X at {hex(X.data_ptr())}")
def one(st, defer=False):
    eng.profile(True); eng.profile_reset()
    eng.run_dense(X, 0, M, out=(st[0], st[1], st[2]), defer=defer)
    eng.synchronize(); torch.cuda.synchronize()
    p = eng.profile_get(); eng.profile(False)
    return p["k_ovo_fused"]["ms"] / p["k_ovo_fused"]["launches"]
for _ in range(8): one(sets[0])
def show(tag, seq, defer=False):
    print(f"{tag:<44}", " ".join(f"{one(sets[i], defer):.3f}" for i in seq))
show("set 0 x 8", [0] * 8)
show("set 1 x 8", [1] * 8)
show("set 2 x 8", [2] * 8)
show("set 3 x 8", [3] * 8)
show("alternating 0 1", [0, 1] * 5)
show("alternating 0 1, deferred", [0, 1] * 5, True)
show("alternating 2 3", [2, 3] * 5)
show("0 0 1 1", [0, 0, 1, 1] * 3)
# untimed back-to-back alternation, the way bench.py runs it, total time over 20 steps
import time
for name, seq in (("alternating 0 1", [0, 1]), ("set 0 only", [0]), ("set 1 only", [1]), ("set 2 only", [2]), ("set 3 only", [3])):
    for st in seq: eng.run_dense(X, 0, M, out=(sets[st][0], sets[st][1], sets[st][2]), defer=True)
    eng.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(20):
        st = sets[seq[k % len(seq)]]
        eng.run_dense(X, 0, M, out=(st[0], st[1], st[2]), defer=True)
    eng.synchronize(); torch.cuda.synchronize()
    print(f"back to back, {name:<18} {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per step")
