#!/bin/bash
# Usage (GPU box): tools/ab_variants.sh <kernel-name prefix> <lib1.so> <lib2.so> ... -- <bench.py args>
# The per-kernel average (rocprofv3 --kernel-trace --stats) of kernels whose name starts with the prefix, for each library put in place of the tree's.
set -u
PFX=$1; shift
LIBS=()
while [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
shift
SO=illico_amd/csrc/libillico_hip.so
cp $SO /tmp/tree.so
export TMPDIR=/tmp
for L in "${LIBS[@]}"; do
  cp $L $SO
  rm -rf gpurun_out/kt_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_ab -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scopes --no-parity --no-c5 --no-extras --no-single-call "$@" > gpurun_out/kt_ab.log 2>&1
  f=$(find gpurun_out/kt_ab -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$PFX" "$L" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith(sys.argv[2]): print(sys.argv[3], r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e6, 3))
PY
done
rm -rf gpurun_out/kt_ab
cp /tmp/tree.so $SO
