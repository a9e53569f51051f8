import sys, time
sys.path.insert(0, "/root/repo")
import torch
from bench import compress, group_container, make_labels, make_matrix
from illico_amd._lib import Engine
N, M, G = 300_000, 8_000, 2_000
dev = torch.device("cuda", 0)
X = make_matrix(torch, N, M, 0.9, 0, dev); csx = compress(torch, X, "csr"); del X
eng = Engine(0); eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_groups(group_container(make_labels(N, G, 0), G, False))
bm = eng.bind_sparse("csr", csx[0], csx[1], csx[2], (N, M))
out = tuple(torch.empty((G, 256), dtype=torch.float64, device=dev) for _ in range(3))
for _ in range(3): bm.run(256, 512, out=out, defer=True); eng.synchronize()
eng.set_option("profile", 1); eng.profile_reset()
bm.run(256, 512, out=out, defer=True); eng.synchronize()
print({k: round(v["ms"], 4) for k, v in eng.profile_get().items()})
eng.set_option("profile", 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): bm.run(256, 512, out=out, defer=True)
eng.synchronize(); torch.cuda.synchronize(); print("per call ms", (time.perf_counter() - t0) / 20 * 1e3)
