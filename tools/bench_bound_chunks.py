#!/usr/bin/env python
"""A bound sparse matrix driven chunk by chunk, as the reference's driver does (asymptotic_wilcoxon.py:213-241: one dispatcher call per
gene chunk; INTEGRATION.md's stub binds the matrix once): total time of the chunk calls against ONE call over every gene.
   python tools/bench_bound_chunks.py [--format csr] [--chunk 256] [--test ovo]"""
import argparse, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from bench import compress, group_container, make_labels, make_matrix
from illico_amd._lib import Engine
ap = argparse.ArgumentParser(); ap.add_argument("--format", default="csr"); ap.add_argument("--chunk", type=int, default=256)
ap.add_argument("--test", default="ovo"); ap.add_argument("--values", default="counts")
ap.add_argument("--ahead", type=int, default=0, help="illico_ctx_set_option bound_ahead_genes for the chunk calls (0: every chunk is a pass of its own)")
a = ap.parse_args()
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, 0.9, 0, dev, values=a.values)
csx = compress(torch, X, a.format); del X
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, a.test == "ovr"))
bm = eng.bind_sparse(a.format, csx[0], csx[1], csx[2], (N, M))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
def whole(): bm.run(0, M, out=out, defer=True); eng.synchronize(); torch.cuda.synchronize()
def chunks(defer):
    eng.set_option("bound_ahead_genes", a.ahead)
    for lb in range(0, M, a.chunk):
        ub = min(M, lb + a.chunk)
        bm.run(lb, ub, out=tuple(t[:, lb:ub] for t in out), defer=defer)
    eng.synchronize(); torch.cuda.synchronize()
    eng.set_option("bound_ahead_genes", 0)
for name, fn in (("one call", whole), (f"{(M + a.chunk - 1) // a.chunk} chunk calls, deferred", lambda: chunks(True)), ("chunk calls, each waited for", lambda: chunks(False))):
    fn(); ts = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{a.format} {a.test} {a.values} ahead={a.ahead}: {name}: {min(ts):.2f} ms")
