"""Tie-heavy NON-INTEGER values, which no float32-count route takes: (a) log1p of raw counts (64 distinct values per gene), (b) counts divided
by a per-cell size factor and log1p'd (scanpy's normalize_total + log1p: the cells that share a total share their values), (c) counts times a
constant.  C2 shape dense and C3 shape CSC / CSR, OVO and OVR; the count-valued and the tie-free continuous passes beside them."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd._lib import Engine
N, M, G = 300000, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 2000
dev = torch.device("cuda:0")
codes = bench.make_labels(N, G, 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
C = bench.make_matrix(torch, N, M, 0.5, 0, dev)                      # counts, half zero
tot = torch.randint(2000, 12000, (N, 1), device=dev, generator=gen).float()   # per-cell totals: ~30 cells share each
kinds = {
    "counts": C,
    "log1p(counts)": torch.log1p(C),
    "counts * 1.37": C * 1.37,
    "log1p(counts / total * 1e4)": torch.log1p(C / tot * 1e4),
    "continuous (no ties)": bench.make_matrix(torch, N, M, 0.5, 0, dev, values="continuous"),
}
for test in ("ovo", "ovr"):
    eng = Engine(0); eng.set_groups(bench.group_container(codes, G, test == "ovr"))
    out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
    for name, X in kinds.items():
        f = lambda: eng.run_dense(X, 0, M, out=out)
        f(); eng.synchronize()
        eng.profile(True); eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(2): f()
        eng.synchronize()
        dt = (time.perf_counter() - t0) / 2 * 1e3
        pr = eng.profile_get(); eng.profile(False)
        top = sorted(((k, round(v["ms"] / 2, 2)) for k, v in pr.items()), key=lambda kv: -kv[1])[:4]
        print(f"{test} dense {name:30s} {dt:9.2f} ms  {top}", flush=True)
    for name in ("counts", "log1p(counts)", "continuous (no ties)"):   # the same values 90 % sparse, as CSC and CSR
        Xs = kinds[name] * (torch.rand((N, M), device=dev, generator=gen) < 0.2)
        for fmt in ("csc", "csr"):
            d, i, p = bench.compress(torch, Xs, fmt)
            f = lambda: eng.run_sparse(fmt, d, i, p, (N, M), 0, M, out=out)
            f(); eng.synchronize()
            eng.profile(True); eng.profile_reset()
            t0 = time.perf_counter()
            for _ in range(2): f()
            eng.synchronize()
            dt = (time.perf_counter() - t0) / 2 * 1e3
            pr = eng.profile_get(); eng.profile(False)
            top = sorted(((k, round(v["ms"] / 2, 2)) for k, v in pr.items()), key=lambda kv: -kv[1])[:4]
            print(f"{test} {fmt:5s} {name:30s} {dt:9.2f} ms  {top}", flush=True)
        del Xs
    eng.close()
