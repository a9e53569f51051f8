#!/usr/bin/env python3
"""Scope (ii) -- host matrix -> host planes -- at C2 shape against the number of host threads that fill the pinned slots
("host_fill_threads"; 16 by default for byte windows) and, optionally, where the process's memory lives (run it under
`numactl --membind / --cpunodebind` yourself: the box's GPU hangs off one socket).

    python tools/host_fill_sweep.py [--threads 8 16 24 32 48 64] [--test ovo|ovr] [--genes 8000]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from illico_amd._lib import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, nargs="+", default=[8, 16, 24, 32, 48, 64])
ap.add_argument("--test", default="ovo")
ap.add_argument("--genes", type=int, default=8000)
ap.add_argument("--cells", type=int, default=300_000)
ap.add_argument("--groups", type=int, default=2000)
ap.add_argument("--mem-node", type=int, default=-1, help="NUMA node whose CPUs first-touch the host matrix (-1: wherever the process runs)")
ap.add_argument("--cpu-node", type=int, default=-1, help="NUMA node the engine's host threads are confined to (-1: no confinement)")
a = ap.parse_args()


def cpus_of(node):
    out = set()
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        lo, _, hi = part.partition("-")
        out.update(range(int(lo), int(hi or lo) + 1))
    return out


dev = torch.device("cuda", 0)
Xd = bench.make_matrix(torch, a.cells, a.genes, 0.5, 0, dev).contiguous()
if a.mem_node >= 0:
    os.sched_setaffinity(0, cpus_of(a.mem_node))
X = np.empty(Xd.shape, dtype=np.float32)          # first touch: by this thread, on the node it runs on
X[...] = 0
X[...] = Xd.cpu().numpy()
del Xd
if a.cpu_node >= 0:
    os.sched_setaffinity(0, cpus_of(a.cpu_node))
elif a.mem_node >= 0:
    os.sched_setaffinity(0, set(range(os.cpu_count())))
print(f"matrix first touched on node {a.mem_node}, engine threads on node {a.cpu_node}", flush=True)
grpc = bench.group_container(bench.make_labels(a.cells, a.groups, 0), a.groups, a.test == "ovr")
eng = Engine(0)
eng.set_groups(grpc)
eng.run_dense(X, 0, a.genes)
for t in a.threads:
    eng.set_option("host_fill_threads", t)
    runs, held = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        held.append(eng.run_dense(X, 0, a.genes))
        runs.append((time.perf_counter() - t0) * 1e3)
    del held
    print(f"{a.test} fill threads {t:3d}: " + " ".join(f"{r:7.1f}" for r in runs) + f"   median {np.median(runs):7.1f} ms", flush=True)
eng.close()
