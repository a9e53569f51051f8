#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_bench.sh <tag> [bench args...]
# Runs bench.py under rocprofv3: one --kernel-trace --stats pass and separate --pmc passes (FETCH_SIZE,
# WRITE_SIZE, SQ counters), then writes gpurun_out/<tag>/summary.json (per-kernel averages per launch).
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scopes --no-parity --no-c5 --no-extras --no-single-call "$@" > $OUT/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scopes --no-parity --no-c5 --no-extras --no-single-call "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scopes --no-parity --no-c5 --no-extras --no-single-call "$@" > $OUT/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scopes --no-parity --no-c5 --no-extras --no-single-call "$@" > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVES --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scopes --no-parity --no-c5 --no-extras --no-single-call "$@" > $OUT/bench_sq2.log 2>&1
python3 tools/summarize_prof.py $OUT
# keep what is judged (summary.json, the --stats table, the bench logs); the raw traces are tens of MB per tag
find $OUT -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/kt $OUT/fetch $OUT/write $OUT/sq $OUT/sq2
