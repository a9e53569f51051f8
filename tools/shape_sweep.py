#!/usr/bin/env python3
"""Shape sweeps through bench.py: ONE table of named shapes (what rounds 3 - 4 kept as twenty shell scripts).

    python tools/shape_sweep.py --list
    python tools/shape_sweep.py no_zeros clusters            # whole sets
    python tools/shape_sweep.py --only c5s_full_ovo tall_dense_cont
    python tools/shape_sweep.py --all --steps 3

Every shape is one `bench.py` process (fresh context, data generated on the device), three timed steps, no CPU baseline, no host
scopes, no C5 leg: the line printed is step time, the five longest kernels of the step, and the bench's own parity leg (16 genes
against the oracle).  Output files go to gpurun_out/s_<tag>.json.  The sweep as a TEST (exact on four genes, within 4x of the
bytes) is tests/test_gpu_shape_sweep.py; this script is for looking at one shape's kernels.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

TALL = "--cells 2000000 --genes 1200 --groups 2000"     # an atlas: the control group (N / 30) has 66 667 cells
WIDE = "--cells 20000 --genes 120000 --groups 100"
CLUS = "--cells 1000000 --genes 2400 --groups 10"       # rank_genes_groups on an atlas: ten clusters of 100 000 cells
SCANPY = "--cells 100000 --genes 30000 --groups 30 --sparsity 0.93"
SCREEN = "--cells 40000 --genes 20000 --groups 300"
TOY = "--cells 500 --genes 20000 --groups 5"

# set -> [(tag, bench.py arguments)]
SETS: dict[str, list[tuple[str, str]]] = {
    "group_counts": [
        ("dense_g50", "--workload c2 --groups 50"),
        ("dense_g10000", "--workload c2 --groups 10000"),
        ("dense_ovr_g50", "--workload c4 --groups 50"),
        ("dense_ovr_g10000", "--workload c4 --groups 10000"),
        ("csc_g50", "--workload c3 --groups 50"),
        ("csc_g5000", "--workload c3 --groups 5000"),
        ("csc_g10000", "--workload c3 --groups 10000"),
        ("csc_ovr_g50", "--workload c3 --test ovr --groups 50"),
        ("csc_ovr_g5000", "--workload c3 --test ovr --groups 5000"),
        ("csr_g300", "--workload c3 --format csr --groups 300"),
        ("csr_g5000", "--workload c3 --format csr --groups 5000"),
        ("csr_ovr_g300", "--workload c3 --format csr --test ovr --groups 300"),
        ("dense_small_n", "--workload c2 --cells 20000 --genes 30000 --groups 100"),
        ("csc_small_n", "--workload c3 --cells 20000 --genes 30000 --groups 100"),
        ("csc_dense50", "--workload c3 --sparsity 0.5"),
        ("csc_sparse99", "--workload c3 --sparsity 0.99"),
    ],
    "continuous_group_counts": [
        ("cont_dense_g10", "--workload c2 --values continuous --groups 10"),
        ("cont_dense_g50", "--workload c2 --values continuous --groups 50"),
        ("cont_dense_g300", "--workload c2 --values continuous --groups 300"),
        ("cont_dense_g2000", "--workload c2 --values continuous"),
        ("cont_dense_g10000", "--workload c2 --values continuous --groups 10000"),
        ("cont_dense_ovr_g50", "--workload c2 --values continuous --test ovr --groups 50"),
        ("cont_dense_ovr_g2000", "--workload c2 --values continuous --test ovr"),
        ("cont_dense_ovr_g10000", "--workload c2 --values continuous --test ovr --groups 10000"),
        ("cont_csc_g50", "--workload c3 --values continuous --groups 50"),
        ("cont_csc_g300", "--workload c3 --values continuous --groups 300"),
        ("cont_csc_g2000", "--workload c3 --values continuous"),
        ("cont_csc_g6000", "--workload c3 --values continuous --groups 6000"),
        ("cont_csc_g10000", "--workload c3 --values continuous --groups 10000"),
        ("cont_csc_ovr_g300", "--workload c3 --values continuous --test ovr --groups 300"),
        ("cont_csc_ovr_g2000", "--workload c3 --values continuous --test ovr"),
        ("cont_csc_ovr_g6000", "--workload c3 --values continuous --test ovr --groups 6000"),
        ("cont_csc_ovr_g10000", "--workload c3 --values continuous --test ovr --groups 10000"),
        ("cont_csr_g300", "--workload c3 --format csr --values continuous --groups 300"),
        ("cont_csr_g2000", "--workload c3 --format csr --values continuous"),
        ("cont_csr_g6000", "--workload c3 --format csr --values continuous --groups 6000"),
        ("cont_csr_ovr_g6000", "--workload c3 --format csr --values continuous --test ovr --groups 6000"),
        ("nb_csc", "--workload c3 --values nb"),
        ("nb_csc_ovr", "--workload c3 --values nb --test ovr"),
        ("nb_csr", "--workload c3 --values nb --format csr"),
    ],
    "csr_shapes": [
        ("csr_s99", "--workload c3 --format csr --sparsity 0.99"),
        ("csr_s99_ovr", "--workload c3 --format csr --sparsity 0.99 --test ovr"),
        ("csr_s70", "--workload c3 --format csr --sparsity 0.7"),
        ("csr_s70_ovr", "--workload c3 --format csr --sparsity 0.7 --test ovr"),
        ("csr_g50", "--workload c3 --format csr --groups 50"),
        ("csr_g50_ovr", "--workload c3 --format csr --groups 50 --test ovr"),
        ("csr_g10000", "--workload c3 --format csr --groups 10000"),
        ("csr_g10000_ovr", "--workload c3 --format csr --groups 10000 --test ovr"),
        ("csr_cont_s99", "--workload c3 --format csr --values continuous --sparsity 0.99"),
        ("csr_cont_g50", "--workload c3 --format csr --values continuous --groups 50"),
        ("csr_cont_g10000_ovr", "--workload c3 --format csr --values continuous --groups 10000 --test ovr"),
        ("csr_nb", "--workload c3 --format csr --values nb"),
        ("csr_nb_ovr", "--workload c3 --format csr --values nb --test ovr"),
        ("csr_mean40", "--workload c3 --format csr --mean-max 40"),
        ("csc_s99", "--workload c3 --sparsity 0.99"),
    ],
    "long_columns": [  # sparse input a fifth and more of whose cells are stored
        ("csc_cont_s70_ovr", "--workload c3 --values continuous --sparsity 0.7 --test ovr"),
        ("csc_cont_s70_ovo", "--workload c3 --values continuous --sparsity 0.7"),
        ("csc_cont_s80_ovr", "--workload c3 --values continuous --sparsity 0.8 --test ovr"),
        ("csc_cont_s80_ovo", "--workload c3 --values continuous --sparsity 0.8"),
        ("csc_cont_s50_ovr", "--workload c3 --values continuous --sparsity 0.5 --test ovr"),
        ("csc_nb_s50", "--workload c3 --values nb --sparsity 0.5"),
        ("csr_cont_s70_ovr", "--workload c3 --format csr --values continuous --sparsity 0.7 --test ovr"),
        ("csr_cont_s70_ovo", "--workload c3 --format csr --values continuous --sparsity 0.7"),
        ("csr_cont_s80_ovr", "--workload c3 --format csr --values continuous --sparsity 0.8 --test ovr"),
        ("csr_cont_s80_ovo", "--workload c3 --format csr --values continuous --sparsity 0.8"),
        ("csr_cont_s85_ovo", "--workload c3 --format csr --values continuous --sparsity 0.85"),
        ("csr_cont_s85_ovr", "--workload c3 --format csr --values continuous --sparsity 0.85 --test ovr"),
        ("dense_cont_s70_ovr", "--workload c2 --values continuous --sparsity 0.7 --test ovr"),
    ],
    "tall": [
        ("tall_dense", f"{TALL} --workload c2"),
        ("tall_dense_ovr", f"{TALL} --workload c4"),
        ("tall_dense_cont", f"{TALL} --workload c2 --values continuous"),
        ("tall_dense_cont_s90", f"{TALL} --workload c2 --values continuous --sparsity 0.9"),
        ("tall_dense_cont_ovr", f"{TALL} --workload c2 --values continuous --test ovr"),
        ("tall_csr", f"{TALL} --workload c3 --format csr"),
        ("tall_csr_ovr", f"{TALL} --workload c3 --format csr --test ovr"),
        ("tall_csc", f"{TALL} --workload c3"),
        ("tall_csc_ovr", f"{TALL} --workload c3 --test ovr"),
        ("tall_csr_cont", f"{TALL} --workload c3 --format csr --values continuous"),
        ("tall_csc_cont", f"{TALL} --workload c3 --values continuous"),
        ("tall_csr_cont_ovr", f"{TALL} --workload c3 --format csr --values continuous --test ovr"),
        ("tall_csc_cont_ovr", f"{TALL} --workload c3 --values continuous --test ovr"),
    ],
    "wide": [
        ("wide_dense", f"{WIDE} --workload c2"),
        ("wide_dense_ovr", f"{WIDE} --workload c4"),
        ("wide_dense_cont", f"{WIDE} --workload c2 --values continuous"),
        ("wide_csr", f"{WIDE} --workload c3 --format csr"),
        ("wide_csc_ovr", f"{WIDE} --workload c3 --test ovr"),
        ("wide_csr_cont_ovr", f"{WIDE} --workload c3 --format csr --values continuous --test ovr"),
    ],
    "tiny_groups": [
        ("tiny_groups_dense", "--groups 30000 --workload c2"),
        ("tiny_groups_dense_ovr", "--groups 30000 --workload c4"),
        ("tiny_groups_csc", "--groups 30000 --workload c3"),
        ("tiny_groups_csr", "--groups 30000 --workload c3 --format csr"),
        ("tiny_groups_dense_cont", "--groups 30000 --workload c2 --values continuous"),
    ],
    "clusters": [
        ("clus_dense_ovr", f"{CLUS} --workload c4"),
        ("clus_dense_ovo", f"{CLUS} --workload c2"),
        ("clus_dense_cont_ovr", f"{CLUS} --workload c2 --values continuous --sparsity 0.9 --test ovr"),
        ("clus_dense_cont_ovo", f"{CLUS} --workload c2 --values continuous --sparsity 0.9"),
        ("clus_dense_cont_ovo_s50", f"{CLUS} --workload c2 --values continuous"),
        ("clus_dense_cont_ovr_s50", f"{CLUS} --workload c2 --values continuous --test ovr"),
        ("clus_dense_cont_ovo_s20", f"{CLUS} --workload c2 --values continuous --sparsity 0.2"),   # runs of 80 000 keys: beyond the 16-bit run lengths
        ("clus_csr_ovr", f"{CLUS} --workload c3 --format csr --test ovr"),
        ("clus_csr_ovo", f"{CLUS} --workload c3 --format csr"),
        ("clus_csc_ovo", f"{CLUS} --workload c3"),
        ("clus_csc_ovr", f"{CLUS} --workload c3 --test ovr"),
        ("clus_csr_cont_ovr", f"{CLUS} --workload c3 --format csr --values continuous --test ovr"),
        ("clus_csr_cont_ovo", f"{CLUS} --workload c3 --format csr --values continuous"),
        ("clus_csc_cont_ovr", f"{CLUS} --workload c3 --values continuous --test ovr"),
        ("clus_csc_cont_ovo", f"{CLUS} --workload c3 --values continuous"),
        ("clus_csr_nb_ovr", f"{CLUS} --workload c3 --format csr --values nb --test ovr"),
    ],
    "everyday": [
        ("scanpy_csr_cont_ovr", f"{SCANPY} --workload c3 --format csr --values continuous --test ovr"),
        ("scanpy_csr_counts_ovr", f"{SCANPY} --workload c3 --format csr --test ovr"),
        ("scanpy_csr_nb_ovr", f"{SCANPY} --workload c3 --format csr --values nb --test ovr"),
        ("scanpy_csc_cont_ovr", f"{SCANPY} --workload c3 --values continuous --test ovr"),
        ("scanpy_dense_cont_ovr", f"{SCANPY} --workload c2 --values continuous --test ovr"),
        ("scanpy_csr_cont_ovo", f"{SCANPY} --workload c3 --format csr --values continuous"),
        ("screen_csr_counts_ovo", f"{SCREEN} --workload c3 --format csr"),
        ("screen_csr_cont_ovo", f"{SCREEN} --workload c3 --format csr --values continuous"),
        ("screen_dense_cont_ovo", f"{SCREEN} --workload c2 --values continuous --sparsity 0.9"),
        ("toy_csr_cont_ovr", f"{TOY} --workload c3 --format csr --values continuous --test ovr"),
        ("toy_dense_counts_ovo", f"{TOY} --workload c2"),
    ],
    "no_zeros": [  # scaled data: continuous values without zeros
        ("c2_full_ovo", "--workload c2 --values continuous --sparsity 0.0"),
        ("c2_full_ovr", "--workload c2 --values continuous --sparsity 0.0 --test ovr"),
        ("c5s_half_ovo", "--workload c5shard --values continuous"),
        ("c5s_full_ovo", "--workload c5shard --values continuous --sparsity 0.0"),
        ("c5s_full_ovr", "--workload c5shard --values continuous --sparsity 0.0 --test ovr"),
        ("c5s_s90_ovo", "--workload c5shard --values continuous --sparsity 0.9"),
    ],
    # the four shapes round 4 left beyond every LDS-resident look-up
    "beyond_lds": [
        ("c5s_full_ovo", "--workload c5shard --values continuous --sparsity 0.0"),
        ("tall_dense_cont", f"{TALL} --workload c2 --values continuous"),
        ("clus_dense_cont_ovo_s50", f"{CLUS} --workload c2 --values continuous"),
        ("csr_cont_f64_ovo", "--workload c3 --format csr --values continuous --dtype f64"),
        ("csr_cont_f32_ovo", "--workload c3 --format csr --values continuous"),
    ],
}
COMMON = "--no-c5 --no-extras --warmup 1 --no-cpu-baseline --no-scopes --no-single-call"


def run_one(tag: str, args: str, steps: int, timeout: int, extra: str) -> str:
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    cmd = [sys.executable, str(ROOT / "bench.py"), *args.split(), *COMMON.split(), "--steps", str(steps), *extra.split()]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return f"{tag:28s} TIMEOUT after {timeout} s"
    (out / f"s_{tag}.json").write_text(r.stdout)
    (out / f"s_{tag}.err").write_text(r.stderr)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        k = d["roofline"]["all_kernels_ms_per_step"]
        top = sorted(k.items(), key=lambda kv: -kv[1])[:5]
        top = ", ".join(f"{n} {v:.2f}" for n, v in top)
        return (f"{tag:28s} {d['ms_per_step']:9.3f} ms  [{top}]  U mismatches {d['parity']['statistic_mismatches']}"
                f"  p err {d['parity']['p_value_max_rel_err']:.1e}")
    except Exception as e:  # noqa: BLE001
        return f"{tag:28s} FAILED ({e}) rc={r.returncode}: {r.stderr[-300:]}"


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("sets", nargs="*", help="names of shape sets")
    ap.add_argument("--only", nargs="+", default=[], help="single shapes by tag")
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--timeout", type=int, default=300)
    ap.add_argument("--extra", default="", help="further bench.py arguments for every shape (e.g. '--engine-option no_ovo_parts=1')")
    a = ap.parse_args()
    if a.list:
        for s, shapes in SETS.items():
            print(s)
            for tag, args in shapes:
                print(f"    {tag:28s} {args}")
        return 0
    todo: list[tuple[str, str]] = []
    for s in (SETS if a.all else a.sets):
        todo += SETS[s]
    by_tag = {tag: args for shapes in SETS.values() for tag, args in shapes}
    todo += [(t, by_tag[t]) for t in a.only]
    seen = set()
    for tag, args in todo:
        if tag in seen:
            continue
        seen.add(tag)
        print(run_one(tag, args, a.steps, a.timeout, a.extra), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
