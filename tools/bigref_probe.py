import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd._lib import Engine
N, M, G = 300000, 8000, 2000
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
n_ref = 60000
codes = np.concatenate([np.zeros(n_ref, dtype=np.int64), 1 + rng.randint(0, G - 1, size=N - n_ref)]); rng.shuffle(codes)
X = bench.make_matrix(torch, N, M, 0.9, 0, dev)
d, i, p = bench.compress(torch, X, "csc")
eng = Engine(0); eng.set_groups(bench.group_container(codes, G, False))
out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
for opt in (0, 1):
    eng.set_option("no_csc_counts_wide", opt)
    eng.run_sparse("csc", d, i, p, (N, M), 0, M, out=out); eng.synchronize()
    eng.profile(True); eng.profile_reset(); t0 = time.perf_counter()
    for _ in range(3): eng.run_sparse("csc", d, i, p, (N, M), 0, M, out=out)
    eng.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    pr = eng.profile_get(); eng.profile(False)
    print("no_csc_counts_wide", opt, round(dt, 3), "ms", sorted(((k, round(v["ms"] / 3, 3)) for k, v in pr.items()), key=lambda kv: -kv[1])[:3], flush=True)
    res = [t.clone() for t in out] if opt == 0 else res
    if opt == 1: print("identical:", all(torch.equal(a, b) for a, b in zip(res, out)))
