import numpy as np, time, torch, os, sys
sys.path.insert(0, '/root/repo')
from illico_amd._lib import Engine
import bench
N, M, G = 300000, 8000, 2000
codes = bench.make_labels(N, G, 0)
g = bench.group_container(codes, G, False)
X = np.random.RandomState(0).poisson(3.0, size=(N, M)).astype(np.float32)
print("cpus", len(os.sched_getaffinity(0)))
eng = Engine(0); eng.set_groups(g)
out = tuple(np.empty((G, M)) for _ in range(3))
for i in range(3):
    t = time.perf_counter(); eng.run_dense(X, 0, M, out=out); print("run_dense host", (time.perf_counter() - t) * 1e3, "ms")
# pinned upload rate
Xp = torch.from_numpy(X).pin_memory()
for i in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); Xd = Xp.cuda(non_blocking=True); torch.cuda.synchronize(); print("pinned H2D", X.nbytes / (time.perf_counter() - t) / 1e9, "GB/s")
t = time.perf_counter(); Y = X.copy(); print("host copy 1 thread", X.nbytes / (time.perf_counter() - t) / 1e9, "GB/s")
dev_out = tuple(torch.empty((G, M), dtype=torch.float64, device="cuda") for _ in range(3))
t = time.perf_counter(); eng.run_dense(Xd, 0, M, out=dev_out); eng.synchronize(); print("device-resident", (time.perf_counter() - t) * 1e3)
t = time.perf_counter(); h = [a.cpu() for a in dev_out]; print("D2H planes pageable", (time.perf_counter() - t) * 1e3, "ms")
