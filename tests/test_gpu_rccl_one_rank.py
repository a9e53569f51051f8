"""The gene-sharded drop-in on RCCL itself (backend "nccl"), with the one rank a single-GPU box allows: a fresh process
initialises the process group on the GPU, `asymptotic_wilcoxon_sharded` computes its planes with the HIP engine, leaves them on
the device and gathers them with `torch.distributed.gather` over RCCL -- the calls the 8-GPU run makes (device-resident
gather lists, asynchronous work handles, stream ordering between the engine and RCCL's stream), checked against the
single-process drop-in call.  The multi-rank logic (sharding, empty ranks, reassembly) is covered on CPU with gloo
(tests/test_distributed_gloo.py)."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pandas as pd
import pytest

ROOT = Path(__file__).resolve().parent.parent

_WORKER = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np, pandas as pd, torch
import torch.distributed as dist
from scipy import sparse
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
from conftest import make_counts, make_labels
from illico_amd import AnnDataLite, asymptotic_wilcoxon
from illico_amd.distributed import asymptotic_wilcoxon_sharded, gather_block_async
X, rng = make_counts(11, 3000, 203, 0.6)
labels = make_labels(rng, 3000, 12, n_ref=300)
fmt, test = {fmt!r}, {test!r}
M = {{"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}}[fmt]
adata = AnnDataLite(M, obs=pd.DataFrame({{"pert": labels}}))
ref = "non-targeting" if test == "ovo" else None
df = asymptotic_wilcoxon_sharded(adata, False, "pert", ref, n_blocks=3, tail="device")   # planes gathered over RCCL, one D2H on rank 0
one = asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", reference=ref)
pd.testing.assert_frame_equal(df, one, check_exact=True)
dfh = asymptotic_wilcoxon_sharded(adata, False, "pert", ref, n_blocks=3)                 # default tail: the engine writes host planes of the shared result
pd.testing.assert_frame_equal(dfh, one, check_exact=True)
# the block gather on its own: device tensors in, device tensors out, asynchronous handle
st = torch.arange(3 * 5 * 7, dtype=torch.float64, device="cuda").reshape(3, 5, 7)
recv = [torch.empty_like(st)]
gather_block_async(st, recv, 0, 1).wait()
torch.cuda.synchronize()
assert torch.equal(recv[0], st)
t = torch.tensor([1.5], device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
df.to_pickle({out!r})
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,test", [("dense", "ovo"), ("csc", "ovr")])
def test_sharded_drop_in_over_rccl_one_rank(tmp_path, fmt, test):
    out = tmp_path / "df.pkl"
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = _WORKER.format(root=str(ROOT), tests=str(ROOT / "tests"), fmt=fmt, test=test, out=str(out))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    df = pd.read_pickle(out)
    assert len(df) == 203 * 12  # one row per (group, gene), the reference group's included (as the reference's frame)
