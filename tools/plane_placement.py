"""How does k_ovo_fused's duration at C2 depend on WHERE the three output planes lie?  (1.655 / 1.860 ms with the
allocation holding them.)  Planes as three separate allocations, and as slices of one buffer with a byte offset between them."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from illico_amd._lib import Engine
N, M, G = 300000, 8000, 2000
dev = torch.device("cuda:0")
X = bench.make_matrix(torch, N, M, 0.5, 0, dev)
codes = bench.make_labels(N, G, 0)
eng = Engine(0); eng.set_groups(bench.group_container(codes, G, False))
plane = G * M  # doubles

def timed(out, tag, reps=12):
    for _ in range(3):
        eng.run_dense(X, 0, M, out=out, defer=True)
    eng.synchronize()
    eng.profile(True); eng.profile_reset()
    for _ in range(reps):
        eng.run_dense(X, 0, M, out=out, defer=True)
    eng.synchronize()
    p = eng.profile_get(); eng.profile(False)
    k = p["k_ovo_fused"]
    print(f"{tag:58s} k_ovo_fused {k['ms'] / k['launches']:.4f} ms   bases mod 2MiB: {[hex(t.data_ptr() % (1 << 21)) for t in out]}  base>>21: {[t.data_ptr() >> 21 for t in out]}", flush=True)

for rep in range(3):
    out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
    timed(out, f"three torch.empty allocations #{rep}")
    keep = out  # keep alive so the next set lands elsewhere
    globals()[f"keep{rep}"] = keep
for pad in [0, 256, 512, 1024, 4096, 8192, 65536, 65536 + 4096, (1 << 20) + 4096, 3 << 20]:
    pd = pad // 8
    buf = torch.empty(3 * (plane + pd) + 1024, dtype=torch.float64, device=dev)
    out = tuple(buf[k * (plane + pd): k * (plane + pd) + plane].view(G, M) for k in range(3))
    timed(out, f"one buffer, planes {pad} bytes apart beyond their size")
    del buf, out
# row pitch: planes with a padded leading dimension
for ldpad in [8, 32, 64, 512]:
    buf = torch.empty(3 * G * (M + ldpad), dtype=torch.float64, device=dev)
    out = tuple(buf[k * G * (M + ldpad): (k + 1) * G * (M + ldpad)].view(G, M + ldpad)[:, :M] for k in range(3))
    timed(out, f"row pitch {M + ldpad} doubles")
    del buf, out
