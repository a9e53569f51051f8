#!/usr/bin/env python
"""Does the C2 kernel's fast / slow mode follow the HBM temperature (temperature-compensated refresh)?  One process runs the
pass continuously for --seconds; every ~0.5 s it prints the mean k_ovo_fused time of the last passes next to rocm-smi's
junction / memory temperatures, power and clocks."""
import argparse, json, subprocess, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=40); ap.add_argument("--idle", type=float, default=0); a = ap.parse_args()
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, 0.5, 0, dev)
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showtemp", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout); c = d[sorted(d)[0]]
        keep = {k: v for k, v in c.items() if any(s in k.lower() for s in ("temperature", "power", "mclk", "sclk", "fclk"))}
        return keep
    except Exception as e:
        return {"err": str(e)[:80]}
print("idle:", smi(), flush=True)
if a.idle: time.sleep(a.idle)
t_end = time.time() + a.seconds
eng.profile(True)
while time.time() < t_end:
    eng.profile_reset()
    t0 = time.time()
    while time.time() - t0 < 0.5:
        for _ in range(20): eng.run_dense(X, 0, M, out=out)
        torch.cuda.synchronize()
    p = eng.profile_get()["k_ovo_fused"]
    s = smi()
    print(f"{p['ms'] / p['launches']:.4f} ms over {p['launches']} passes  {s}", flush=True)
