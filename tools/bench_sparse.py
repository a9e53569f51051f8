#!/usr/bin/env python
"""C3-style measurement: K562-shaped CSC/CSR input (device-resident), OVO or OVR, one engine pass per step."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import group_container, make_labels, make_matrix
from illico_amd._lib import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--fmt", default="csc"); ap.add_argument("--test", default="ovo")
ap.add_argument("--cells", type=int, default=300_000); ap.add_argument("--genes", type=int, default=8_000)
ap.add_argument("--groups", type=int, default=2_000); ap.add_argument("--sparsity", type=float, default=0.9)
ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--values", default="counts", choices=["counts", "continuous"])
ap.add_argument("--engine-option", action="append", default=[], help="key=value, repeatable")
a = ap.parse_args()
dev = torch.device("cuda", 0)
N, M, G = a.cells, a.genes, a.groups
grpc = group_container(make_labels(N, G, 0), G, a.test == "ovr")
X = make_matrix(torch, N, M, a.sparsity, 0, dev, continuous=a.values == "continuous")
# compressed arrays built block by block on the device (CSC = CSR of X^T)
datas, idxs, cnts = [], [], []
if a.fmt == "csc":
    for j0 in range(0, M, 256):
        Xb = X[:, j0:j0 + 256].t().contiguous()
        nz = Xb != 0
        cnts.append(nz.sum(1)); idxs.append(nz.nonzero()[:, 1].to(torch.int32)); datas.append(Xb[nz])
else:
    for r0 in range(0, N, 16384):
        Xb = X[r0:r0 + 16384]
        nz = Xb != 0
        cnts.append(nz.sum(1)); idxs.append(nz.nonzero()[:, 1].to(torch.int32)); datas.append(Xb[nz])
data, indices = torch.cat(datas).contiguous(), torch.cat(idxs).contiguous()
indptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cat(cnts).cumsum(0)]).to(torch.int32).contiguous()
nnz = int(data.numel()); del X, datas, idxs, cnts
torch.cuda.synchronize()
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream); eng.set_groups(grpc)
for kv in a.engine_option:
    k, v = kv.split("="); eng.set_option(k, int(v))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
run = lambda: eng.run_sparse(a.fmt, data, indices, indptr, (N, M), 0, M, out=out)
for _ in range(a.warmup): run()
torch.cuda.synchronize(); eng.profile(True); eng.profile_reset(); t0 = time.perf_counter()
for _ in range(a.steps): run()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
prof = eng.profile_get()
alg = nnz * 8 + (M + 1) * 4 + 4 * N + 24 * G * M
print(json.dumps({"workload": f"{a.fmt} {N}x{M}x{G} {a.test} sparsity {a.sparsity} {a.values} {' '.join(a.engine_option)}", "nnz": nnz, "ms_per_step": round(dt * 1e3, 3),
                  "tests_per_s": round(G * M / dt, 1), "algorithmic_bytes": alg, "achieved_GBs_pipeline": round(alg / dt / 1e9, 1),
                  "kernels_ms_per_step": {k: round(v["ms"] / a.steps, 3) for k, v in prof.items()}}))
