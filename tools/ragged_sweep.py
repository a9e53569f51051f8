"""Clusters of very different sizes (a Dirichlet draw: from a few hundred to tens of thousands of cells) on a scanpy-sized matrix,
100 000 cells x 8192 genes x 30 groups, against the same matrix with equal groups: dense counts / continuous, CSR continuous; OVO / OVR."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd._lib import Engine
N, M, G = 100000, 8192, 30
dev = torch.device("cuda:0")
rng = np.random.RandomState(4)
sizes = rng.multinomial(N - G * 50, rng.dirichlet(np.ones(G) * 0.5)) + 50
ragged = np.repeat(np.arange(G), sizes); rng.shuffle(ragged)
equal = bench.make_labels(N, G, 0)
print("ragged sizes:", sorted(sizes.tolist())[:3], "...", sorted(sizes.tolist())[-3:])
mats = {"counts": bench.make_matrix(torch, N, M, 0.5, 0, dev), "continuous": bench.make_matrix(torch, N, M, 0.9, 0, dev, values="continuous")}
csr = bench.compress(torch, mats["continuous"], "csr")
sp_counts = bench.make_matrix(torch, N, M, 0.9, 0, dev)
csr_c = bench.compress(torch, sp_counts, "csr")
csc_c = bench.compress(torch, sp_counts, "csc")
csc = bench.compress(torch, mats["continuous"], "csc")
del sp_counts
for test in ("ovo", "ovr"):
    for lname, codes in (("equal", equal), ("ragged", ragged)):
        eng = Engine(0)
        for kv in sys.argv[1:]:
            k, v = kv.split('='); eng.set_option(k, int(v))
        eng.set_groups(bench.group_container(codes.astype(np.int64), G, test == "ovr"))
        out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
        runs = {"dense counts": lambda: eng.run_dense(mats["counts"], 0, M, out=out), "dense continuous": lambda: eng.run_dense(mats["continuous"], 0, M, out=out),
                "csr continuous": lambda: eng.run_sparse("csr", csr[0], csr[1], csr[2], (N, M), 0, M, out=out),
                "csc continuous": lambda: eng.run_sparse("csc", csc[0], csc[1], csc[2], (N, M), 0, M, out=out),
                "csr counts": lambda: eng.run_sparse("csr", csr_c[0], csr_c[1], csr_c[2], (N, M), 0, M, out=out),
                "csc counts": lambda: eng.run_sparse("csc", csc_c[0], csc_c[1], csc_c[2], (N, M), 0, M, out=out)}
        for name, f in runs.items():
            f(); eng.synchronize()
            eng.profile(True); eng.profile_reset()
            t0 = time.perf_counter()
            for _ in range(3): f()
            eng.synchronize()
            dt = (time.perf_counter() - t0) / 3 * 1e3
            pr = eng.profile_get(); eng.profile(False)
            top = sorted(((k, round(v["ms"] / 3, 2)) for k, v in pr.items()), key=lambda kv: -kv[1])[:4]
            print(f"{test} {lname:7s} {name:18s} {dt:8.2f} ms  {top}", flush=True)
        eng.close()
