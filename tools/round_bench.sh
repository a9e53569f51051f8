#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/round_bench.sh <round tag, e.g. r03> [part: b1 b2 b3 b4 b5 p1 p2 p3 p4 p5 chunks; default all]
# (a gpurun call lasts 20 minutes at most: the whole script does not fit one)
# The bench lines and rocprofv3 summaries a round commits under profiles/: every line into gpurun_out/<tag>_bench_*.json,
# every profile into gpurun_out/<tag>_<workload>/ (tools/profile_bench.sh).  Prints a progress line per step.
set -u
T=$1
PART=${2:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
O=gpurun_out
mkdir -p $O
b() { name=$1; shift; python3 bench.py "$@" > $O/${T}_bench_$name.json 2> $O/${T}_bench_$name.err; echo "bench $name rc=$?"; }
want b1 && b c2 
want b1 && b c3 --workload c3 --no-c5 --no-extras
want b1 && b c3_ovr --workload c3 --test ovr --no-c5 --no-extras
want b1 && b c4 --workload c4 --no-c5 --no-extras
want b1 && b c5shard --workload c5shard --no-c5 --no-extras
want b1 && b c5_one_gpu --workload c5 --no-c5 --no-extras --steps 5 --warmup 1
want b1 && b c2_nb --workload c2 --values nb --no-c5 --no-extras
want b2 && b c2_nb_ovr --workload c2 --values nb --test ovr --no-c5 --no-extras
want b2 && b c2_cont_ovo --workload c2 --values continuous --no-c5 --no-extras --steps 10
want b2 && b c2_cont_ovr --workload c2 --values continuous --test ovr --no-c5 --no-extras --steps 10
want b2 && b c3_cont_ovo --workload c3 --values continuous --no-c5 --no-extras --steps 10
want b2 && b c3_cont_ovr --workload c3 --values continuous --test ovr --no-c5 --no-extras --steps 10
want b2 && b c3_csr --workload c3 --format csr --no-c5 --no-extras --steps 10
want b2 && b c3_csr_ovr --workload c3 --format csr --test ovr --no-c5 --no-extras --steps 10
want b3 && b c3_csr_cont_ovo --workload c3 --format csr --values continuous --no-c5 --no-extras --steps 10
want b3 && b c3_csr_cont_ovr --workload c3 --format csr --values continuous --test ovr --no-c5 --no-extras --steps 10
want b3 && b c2_cont_g50 --workload c2 --values continuous --groups 50 --no-c5 --no-extras --steps 5
want b3 && b c3_cont_g300 --workload c3 --values continuous --groups 300 --no-c5 --no-extras --steps 5
want b3 && b c3_cont_ovr_g6000 --workload c3 --values continuous --test ovr --groups 6000 --no-c5 --no-extras --steps 5
# continuous OVO beyond every LDS-resident look-up (round 5): value-range parts of the reference, runs dealt through HBM, float64 sparse
want b4 && b c5shard_cont_nozeros --workload c5shard --values continuous --sparsity 0.0 --no-c5 --no-extras --steps 5
want b4 && b tall_cont_ovo --cells 2000000 --genes 1200 --groups 2000 --workload c2 --values continuous --no-c5 --no-extras --steps 5
want b4 && b clusters_cont_ovo --cells 1000000 --genes 2400 --groups 10 --workload c2 --values continuous --no-c5 --no-extras --steps 5
want b4 && b c3_csr_cont_f64_ovo --workload c3 --format csr --values continuous --dtype f64 --no-c5 --no-extras --steps 10
want b4 && b wide_cont_ovo --cells 20000 --genes 120000 --groups 100 --workload c2 --values continuous --no-c5 --no-extras --steps 5
want b4 && b c3_g30000 --workload c3 --groups 30000 --no-c5 --no-extras --steps 5
# atlas shapes (round 5, second half): clusters of 100 000 cells and columns of two million cells, every test and value kind
want b5 && b clusters_counts_ovo --cells 1000000 --genes 2400 --groups 10 --workload c2 --no-c5 --no-extras --steps 10
want b5 && b clusters_counts_ovr --cells 1000000 --genes 2400 --groups 10 --workload c4 --no-c5 --no-extras --steps 10
want b5 && b clusters_cont_ovr_s90 --cells 1000000 --genes 2400 --groups 10 --workload c2 --values continuous --sparsity 0.9 --test ovr --no-c5 --no-extras --steps 5
want b5 && b clusters_cont_ovo_s90 --cells 1000000 --genes 2400 --groups 10 --workload c2 --values continuous --sparsity 0.9 --no-c5 --no-extras --steps 5
want b5 && b clusters_cont_ovr --cells 1000000 --genes 2400 --groups 10 --workload c2 --values continuous --test ovr --no-c5 --no-extras --steps 5
want b5 && b tall_cont_ovr --cells 2000000 --genes 1200 --groups 2000 --workload c2 --values continuous --test ovr --no-c5 --no-extras --steps 5
want b5 && b tall_counts_ovr --cells 2000000 --genes 1200 --groups 2000 --workload c4 --no-c5 --no-extras --steps 10
want b5 && b wide_csr_cont_ovr --cells 20000 --genes 120000 --groups 100 --workload c3 --format csr --values continuous --test ovr --no-c5 --no-extras --steps 5
want b5 && b scanpy_csr_cont_ovr --cells 100000 --genes 30000 --groups 30 --sparsity 0.93 --workload c3 --format csr --values continuous --test ovr --no-c5 --no-extras --steps 5
p() { name=$1; shift; bash tools/profile_bench.sh ${T}_$name "$@" > $O/prof_$name.log 2>&1; echo "profile $name rc=$?"; }
want p1 && p c2 --workload c2
want p1 && p c3 --workload c3
want p1 && p c3_ovr --workload c3 --test ovr
want p1 && p c4 --workload c4
want p1 && p c5shard --workload c5shard
want p2 && p c2_nb --workload c2 --values nb
want p2 && p c2_cont_ovr --workload c2 --values continuous --test ovr
want p2 && p c3_cont_ovr --workload c3 --values continuous --test ovr
want p2 && p c2_nb_ovr --workload c2 --values nb --test ovr
want p2 && p c3_csr --workload c3 --format csr
want p3 && p c3_csr_ovr --workload c3 --format csr --test ovr
want p3 && p c2_cont_ovo --workload c2 --values continuous
want p3 && p c3_csr_cont_ovr --workload c3 --format csr --values continuous --test ovr
want p3 && p c3_cont_ovo --workload c3 --values continuous
want p4 && p c5shard_cont_nozeros --workload c5shard --values continuous --sparsity 0.0
want p4 && p clusters_cont_ovo --cells 1000000 --genes 2400 --groups 10 --workload c2 --values continuous
want p5 && p clusters_counts_ovo --cells 1000000 --genes 2400 --groups 10 --workload c2
want p5 && p clusters_cont_ovr_s90 --cells 1000000 --genes 2400 --groups 10 --workload c2 --values continuous --sparsity 0.9 --test ovr
# the reference driver's chunking on a bound CSR matrix, with and without the windows computed ahead
want chunks && { for a in 0 2048; do for t in ovo ovr; do python3 tools/bench_bound_chunks.py --ahead $a --test $t; done; done > $O/${T}_bound_chunks.txt 2>&1; echo "bound chunks rc=$?"; }
exit 0
