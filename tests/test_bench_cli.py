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
    a = bench.parse(["--workload", "c5"])
    assert (a.cells, a.genes, a.groups, a.scaling) == (1_000_000, 30_000, 5_000, "strong")   # configs[4] whole, genes split over the ranks
    assert (a.c5_cells, a.c5_genes, a.c5_groups) == (1_000_000, 30_000, 5_000) and not a.no_c5  # ... and in every default line (c5_strong)


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
    """`python bench.py --gpus 2` with no launcher: bench.py starts the ranks, they rendezvous, split ONE workload's genes
    (strong scaling, the default), run the timed steps WITH the gather inside them and rank 0 prints one line with n_gpus = 2,
    value = tests / (pass + gather), the pass-only figure beside it and the c5_strong object (BASELINE configs[4], here at a
    reduced size).  (gloo + one shared GPU: the box has a single GPU; RCCL itself runs in the driver's 8-GPU tier.)"""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--steps", "2", "--warmup", "1", "--cells", "20000", "--genes", "256", "--groups", "50", "--no-cpu-baseline", "--no-scopes",
              "--c5-cells", "30000", "--c5-genes", "192", "--c5-groups", "40", "--c5-steps", "2"]
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", *common],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["genes_total"] == 256 and two["config"]["genes_per_gpu"] == 128
    # (a) + (b): the headline includes the gather, the pass-only figure is secondary
    assert two["final_gather"]["in_timed_step"] is True and two["final_gather"]["bytes_into_rank0_per_step"] == 24 * 50 * 128
    assert two["value"] == pytest.approx(50 * 256 / (two["ms_per_step"] * 1e-3), rel=1e-3)
    assert two["pass_only"]["ms_per_step"] > 0 and two["pass_only"]["tests_per_s"] == pytest.approx(50 * 256 / (two["pass_only"]["ms_per_step"] * 1e-3), rel=1e-3)
    assert two["parity"]["statistic_mismatches"] == 0 and two["parity"]["p_value_max_rel_err"] <= 1e-12 and len(two["parity"]["genes_checked"]) == 16
    # a single call: the gathers of its own blocks only, at least four blocks, its own N = 1 reference and speed-up in the same line
    sc = two["single_call"]
    assert sc["blocks"] >= 4 and sc["ms_single_call"] > 0 and len(sc["runs_ms"]) == 5 and sc["ms_gather_alone"] > 0 and sc["gather_into_rank0_GBs"] > 0
    assert sc["n1_reference"]["genes"] == 256 and sc["n1_reference"]["ms_single_call"] > 0
    assert sc["speedup_vs_n1"] == pytest.approx(sc["n1_reference"]["ms_single_call"] / sc["ms_single_call"], rel=1e-2)
    assert two["steady_state"]["ms_per_step"] == two["ms_per_step"]
    # ... and the drop-in scope of the same call: every rank's planes into its own columns of one shared host result
    assert sc["ms_to_host"] > 0 and sc["to_host"]["bytes_to_host_total"] == 24 * 50 * 256 and sc["to_host"]["bytes_to_host_per_rank"] == 24 * 50 * 128
    # (c): configs[4] in the same line
    c5 = two["c5_strong"]
    assert c5["single_call"]["blocks"] >= 4 and c5["single_call"]["speedup_vs_n1"] > 0 and c5["single_call"]["n1_reference"]["genes"] == 192
    assert c5["single_call"]["ms_to_host"] > 0 and c5["single_call"]["to_host"]["bytes_to_host_total"] == 24 * 40 * 192
    assert c5["genes_total"] == 192 and c5["genes_per_gpu"] == 96 and c5["cells"] == 30000 and c5["groups"] == 40 and c5["scaling"] == "strong"
    assert c5["ms_pass"] > 0 and c5["ms_gather_alone"] > 0 and c5["ms_pass_plus_gather"] > 0 and c5["bytes_into_rank0"] == 24 * 40 * 96
    assert 0 < c5["roofline"]["frac"] < 1 and c5["parity"]["statistic_mismatches"] == 0
    # one GPU: the same workload, the same matrix (blocks seeded by (seed, block number)), no gather
    r1 = subprocess.run([sys.executable, str(ROOT / "bench.py"), *common], capture_output=True, text=True, env=env, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and one["roofline"]["kernel"] == two["roofline"]["kernel"]
    assert one["config"]["genes_total"] == 256 and "final_gather" not in one
    assert one["c5_strong"]["genes_per_gpu"] == 192 and one["c5_strong"]["bytes_into_rank0"] == 0 and one["c5_strong"]["ms_gather_alone"] == 0
    assert one["single_call"]["blocks"] == 1 and one["single_call"]["ms_single_call"] > 0 and "speedup_vs_n1" not in one["single_call"]
    assert one["single_call"]["ms_to_host"] > 0 and one["c5_strong"]["single_call"]["ms_to_host"] > 0
    # the other single-GPU BASELINE configs and one continuous line ride in the same (N = 1) line, at the headline's shape
    assert all(k not in two for k in ("c3", "c3_csr", "c4", "c2_continuous_ovr"))
    for tag, fmt, test, values in (("c3", "csc", "ovo", "counts"), ("c3_csr", "csr", "ovo", "counts"), ("c4", "dense", "ovr", "counts"),
                                   ("c2_continuous_ovr", "dense", "ovr", "continuous")):
        x = one[tag]
        assert (x["format"], x["test"], x["values"]) == (fmt, test, values), x
        assert x["ms_per_step"] > 0 and x["tests_per_s"] == pytest.approx(50 * 256 / (x["ms_per_step"] * 1e-3), rel=2e-2)   # (ms_per_step is rounded to 0.1 us)
        rf = x["roofline"]
        assert rf["bound"] == "hbm" and rf["kernel"] and 0 < rf["frac"] < 1 and 0 < rf["pipeline_frac"] < 1 and "traffic" in rf and rf["avg_launch_ms"] > 0
        assert x["parity"]["statistic_mismatches"] == 0 and x["parity"]["p_value_max_rel_err"] <= 1e-12 and x["parity"]["fold_change_max_rel_err"] <= 1e-12
        assert x["cpu_baseline"] is None   # (--no-cpu-baseline in `common`)
    # ... with their CPU baselines when the headline has one
    r3 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--cells", "20000", "--genes", "256", "--groups", "50",
                         "--no-scopes", "--no-c5", "--no-single-call", "--cpu-seconds", "0.3", "--extras-cpu-seconds", "0.2", "--extras-steps", "2"],
                        capture_output=True, text=True, env=env, timeout=900)
    assert r3.returncode == 0, r3.stderr[-2000:]
    ex = json.loads([l for l in r3.stdout.splitlines() if l.startswith("{")][0])
    for tag in ("c3", "c3_csr", "c4", "c2_continuous_ovr"):
        cb = ex[tag]["cpu_baseline"]
        assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["at_8_threads"]["value"] > 0 and "sample" in cb
    # weak scaling on request: every rank a full shard; the gather outside the step on request
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--scaling", "weak",
                         "--no-gather-in-step", "--no-c5", *common], capture_output=True, text=True, env=env, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    wk = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert wk["scaling"] == "weak" and wk["config"]["genes_total"] == 512 and wk["config"]["genes_per_gpu"] == 256
    assert wk["final_gather"]["in_timed_step"] is False and wk["final_gather"]["ms"] >= 0 and "c5_strong" not in wk


@pytest.mark.gpu
def test_matrix_blocks_do_not_depend_on_the_sharding():
    """A rank's gene range holds exactly the columns a single GPU would have generated (strong scaling = the same problem)."""
    import torch
    sys.path.insert(0, str(ROOT))
    import bench
    dev = torch.device("cuda", 0)
    whole = bench.make_matrix(torch, 500, 700, 0.5, 3, dev)
    for lb, ub in ((0, 350), (350, 700), (100, 613)):
        part = bench.make_matrix(torch, 500, ub - lb, 0.5, 3, dev, gene_lb=lb)
        assert torch.equal(part, whole[:, lb:ub])
    nb = bench.make_matrix(torch, 20000, 512, 0.5, 1, dev, values="nb")
    mx = nb.max(dim=0).values
    assert 0.05 < float((mx > 63).float().mean()) < 0.45 and float((mx > 255).float().mean()) > 0.005
