"""Step time of the count-valued dense and CSC passes for every value dtype (float32 / float64 / int32 / int64) and of the float32 pass
with is_log1p=True, at 300k x 4000 x 2000 -- a look for cliffs the float32 benchmarks cannot show."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd._lib import Engine
N, M, G = 300000, 4000, 2000
dev = torch.device("cuda:0")
X32 = bench.make_matrix(torch, N, M, 0.9, 0, dev)
codes = bench.make_labels(N, G, 0)
for test in ("ovo", "ovr"):
    eng = Engine(0); eng.set_groups(bench.group_container(codes, G, test == "ovr"))
    out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
    def timed(tag, f, reps=5):
        f(); eng.synchronize()
        eng.profile(True); eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        p = eng.profile_get(); eng.profile(False)
        top = sorted(((k, round(v["ms"] / reps, 3)) for k, v in p.items()), key=lambda kv: -kv[1])[:3]
        print(f"{test} {tag:28s} {dt:8.3f} ms  {top}", flush=True)
    for dt in (torch.float32, torch.float64, torch.int32, torch.int64):
        X = X32.to(dt).contiguous()
        timed(f"dense {str(dt)[6:]}", lambda: eng.run_dense(X, 0, M, out=out))
        d, i, p = bench.compress(torch, X32, "csc")
        d = d.to(dt)
        timed(f"csc {str(dt)[6:]}", lambda: eng.run_sparse("csc", d, i, p, (N, M), 0, M, out=out))
        ii, pp = i.to(torch.int64), p.to(torch.int64)
        if dt == torch.float32:
            timed("csc float32, int64 indices", lambda: eng.run_sparse("csc", d, ii, pp, (N, M), 0, M, out=out))
            timed("dense float32 is_log1p", lambda: eng.run_dense(X, 0, M, out=out, is_log1p=True), reps=2)
            timed("csc float32 is_log1p", lambda: eng.run_sparse("csc", d, i, p, (N, M), 0, M, out=out, is_log1p=True), reps=2)
        del X, d
