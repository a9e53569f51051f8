#!/bin/bash
# Usage (on the GPU box): tools/ab_lib.sh <base .so> <out prefix> -- <bench.py args>
# A/B of two builds of the library on ONE box: the bench line with the library in the tree, then with <base .so> put in its place
# (the copy on the box is scratch), then the tree's again (drift check).
set -u
BASE=$1; OUT=$2; shift 3
SO=illico_amd/csrc/libillico_hip.so
cp $SO /tmp/new.so
python3 bench.py "$@" > ${OUT}_new.json 2> ${OUT}_new.err || exit 1
cp $BASE $SO
python3 bench.py "$@" > ${OUT}_base.json 2> ${OUT}_base.err || exit 1
cp /tmp/new.so $SO
python3 bench.py "$@" > ${OUT}_new2.json 2> ${OUT}_new2.err || exit 1
python3 - <<PY
import json
for t in ("new", "base", "new2"):
    d = json.loads(open("${OUT}_%s.json" % t).read().strip().splitlines()[-1])
    print("${OUT}", t, d["ms_per_step"], d["roofline"].get("all_kernels_ms_per_step"))
PY
