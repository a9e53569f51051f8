#!/usr/bin/env python
"""Where do the fast / slow modes of the C2 step come from?  One process, one engine, one X; the three output planes are
carved out of one pool at chosen base offsets, plane strides and row pitches, and k_ovo_fused is timed (HIP events) for
each placement.  Usage: python tools/mode_probe2.py [--reps 6]"""
import argparse, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=6); a = ap.parse_args()
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, 0.5, 0, dev)
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
pool = torch.empty(3 * (G * (M + 512) * 8) + (512 << 20), dtype=torch.uint8, device=dev)
base = pool.data_ptr()
print(f"pool at {hex(base)}")
def planes(off, stride, ld):
    out = []
    for k in range(3):
        b = off + k * stride
        assert b % 8 == 0 and b + G * ld * 8 <= pool.numel()
        out.append(pool[b: b + G * ld * 8].view(torch.float64).view(G, ld)[:, :M])
    return tuple(out)
def timed(out):
    eng.run_dense(X, 0, M, out=out); torch.cuda.synchronize()
    eng.profile(True); eng.profile_reset()
    for _ in range(a.reps): eng.run_dense(X, 0, M, out=out)
    torch.cuda.synchronize()
    p = eng.profile_get(); eng.profile(False)
    return p["k_ovo_fused"]["ms"] / p["k_ovo_fused"]["launches"]
S0 = G * M * 8
al = (-base) % (2 << 20)   # offset that makes the first plane 2 MiB aligned
print("A. base offset (planes contiguous, ld = M)")
for off in (0, al, al + 256, al + 4096, al + (64 << 10), al + (1 << 20), al + (2 << 20), al + (6 << 20), al + (32 << 20), al + (64 << 20) + 4096):
    print(f"   off {off:>10}  ({hex((base + off) & 0xFFFFFFF)}): {timed(planes(off, S0, M)):.4f} ms")
print("B. plane stride (first plane 2 MiB aligned)")
for pad in (0, 256, 1024, 4096, 16384, 65536, 1 << 20, (2 << 20) - S0 % (2 << 20), (2 << 20) - S0 % (2 << 20) + 4096, (2 << 20) - S0 % (2 << 20) + (1 << 20)):
    print(f"   stride S0 + {pad:>8}: {timed(planes(al, S0 + pad, M)):.4f} ms")
print("C. row pitch (out_ld)")
for ld in (M, M + 8, M + 16, M + 32, M + 64, M + 128, M + 192, M + 256):
    print(f"   out_ld {ld}: {timed(planes(al, G * ld * 8, ld)):.4f} ms")
print("D. repeat of the first placement (drift check)")
print(f"   {timed(planes(0, S0, M)):.4f} ms")
