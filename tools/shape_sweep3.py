"""A few more shapes: a large OVO reference on dense and CSR input, two million cells."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
import oracle
from illico_amd._lib import Engine
dev = torch.device("cuda:0")
def go(tag, N, M, G, n_ref, fmt, ovr=False, sparsity=0.5):
    rng = np.random.RandomState(0)
    codes = np.concatenate([np.zeros(n_ref, dtype=np.int64), 1 + rng.randint(0, G - 1, size=N - n_ref)]); rng.shuffle(codes)
    X = bench.make_matrix(torch, N, M, sparsity, 0, dev)
    grpc = bench.group_container(codes, G, ovr)
    eng = Engine(0); eng.set_groups(grpc)
    out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
    if fmt == "dense":
        f = lambda: eng.run_dense(X, 0, M, out=out)
    else:
        d, i, p = bench.compress(torch, X, fmt)
        f = lambda: eng.run_sparse(fmt, d, i, p, (N, M), 0, M, out=out)
    f(); eng.synchronize()
    eng.profile(True); eng.profile_reset(); t0 = time.perf_counter()
    for _ in range(3): f()
    eng.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
    pr = eng.profile_get(); eng.profile(False)
    # parity on 4 genes
    genes = [0, M // 3, 2 * M // 3, M - 1]
    Xh = X[:, genes].cpu().numpy().astype(np.float64)
    want = oracle.run(Xh, grpc)
    got_u = out[1][:, genes].cpu().numpy()
    mask = np.ones(G, bool)
    if not ovr: mask[0] = False
    mism = int((got_u[mask] != want[1][mask]).sum())
    print(f"{tag:34s} {dt:8.3f} ms  {sorted(((k, round(v['ms'] / 3, 3)) for k, v in pr.items()), key=lambda kv: -kv[1])[:3]}  U mismatches on 4 genes: {mism}", flush=True)
    del X, out
go("dense OVO, reference 100k", 300000, 8000, 2000, 100000, "dense")
go("csr OVO, reference 100k", 300000, 8000, 2000, 100000, "csr", sparsity=0.9)
go("csc OVO, reference 100k", 300000, 8000, 2000, 100000, "csc", sparsity=0.9)
go("dense OVO, reference 3 cells", 300000, 8000, 2000, 3, "dense")
go("csc OVO, reference 3 cells", 300000, 8000, 2000, 3, "csc", sparsity=0.9)
go("dense OVR, 2M cells x 2000", 2000000, 2000, 2000, 60000, "dense", ovr=True)
go("dense OVO, 2M cells x 2000", 2000000, 2000, 2000, 60000, "dense")
go("csc OVR, 2M cells x 2000", 2000000, 2000, 2000, 60000, "csc", ovr=True, sparsity=0.9)
