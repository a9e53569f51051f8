"""is_log1p=True against False on continuous (log1p-normalised-like) values: the flag only changes the fold change's value sums
(expm1 of every value), so the two should cost about the same.  C2 shape dense, C3 shape CSC."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd._lib import Engine
N, M, G = 300000, 8000, 2000
dev = torch.device("cuda:0")
codes = bench.make_labels(N, G, 0)
Xd = bench.make_matrix(torch, N, M, 0.5, 0, dev, values="continuous")
Xs = bench.make_matrix(torch, N, M, 0.9, 0, dev, values="continuous")
d, i, p = bench.compress(torch, Xs, "csc")
del Xs
for test in ("ovo", "ovr"):
    eng = Engine(0); eng.set_groups(bench.group_container(codes, G, test == "ovr"))
    out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
    def timed(tag, f, reps=3):
        f(); eng.synchronize()
        eng.profile(True); eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        pr = eng.profile_get(); eng.profile(False)
        top = sorted(((k, round(v["ms"] / reps, 3)) for k, v in pr.items()), key=lambda kv: -kv[1])[:4]
        print(f"{test} {tag:28s} {dt:8.3f} ms  {top}", flush=True)
    for lg in (False, True):
        timed(f"dense continuous log1p={lg}", lambda: eng.run_dense(Xd, 0, M, out=out, is_log1p=lg))
        timed(f"csc continuous log1p={lg}", lambda: eng.run_sparse("csc", d, i, p, (N, M), 0, M, out=out, is_log1p=lg))
