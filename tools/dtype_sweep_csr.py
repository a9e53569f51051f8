"""Step time of the CSR passes (counts: the group-major pass; continuous: transposition + CSC kernels) for float64 values and int64 indices at
C3 shape -- a look for cliffs the float32 / int32 benchmarks cannot show (tools/dtype_sweep.py does the same for dense and CSC)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd._lib import Engine
N, M, G = 300000, 8000, 2000
dev = torch.device("cuda:0")
codes = bench.make_labels(N, G, 0)
for values in ("counts", "continuous"):
    X = bench.make_matrix(torch, N, M, 0.9, 0, dev, values=values)
    d32, i32, p32 = bench.compress(torch, X, "csr"); del X
    for test in ("ovo", "ovr"):
        eng = Engine(0); eng.set_groups(bench.group_container(codes, G, test == "ovr"))
        out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
        for vdt, idt in ((torch.float32, torch.int32), (torch.float64, torch.int32), (torch.float32, torch.int64), (torch.float64, torch.int64)):
            d, i, p = d32.to(vdt), i32.to(idt), p32.to(idt)
            def f(): eng.run_sparse("csr", d, i, p, (N, M), 0, M, out=out, defer=True); eng.synchronize()
            f()
            eng.profile(True); eng.profile_reset()
            t0 = time.perf_counter()
            for _ in range(3): f()
            dt = (time.perf_counter() - t0) / 3 * 1e3
            pr = eng.profile_get(); eng.profile(False)
            top = sorted(((k, round(v["ms"] / 3, 3)) for k, v in pr.items()), key=lambda kv: -kv[1])[:3]
            print(f"csr {values} {test} {str(vdt)[6:]:8s} {str(idt)[6:]:6s} {dt:8.3f} ms  {top}", flush=True)
            del d, i, p
