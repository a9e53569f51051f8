#!/bin/bash
# Usage (GPU box): tools/mode_pmc.sh <n>  -- n processes of tools/mode_probe2.py, each under rocprofv3 with translation-cache counters;
# prints per process: k_ovo_fused's HIP-event time (from the script) and the counters of its last dispatch.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mode_pmc
mkdir -p $OUT
cd $R
for i in $(seq 1 ${1:-4}); do
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p$i -- python3 tools/mode_probe2.py > $OUT/p$i.log 2>&1
  grep k_ovo_fused $OUT/p$i.log
  python3 - $OUT/p$i <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "k_ovo_fused" in r["Kernel_Name"] and "Lb1" not in r["Kernel_Name"][-40:]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:40]].add(r["Dispatch_Id"])
    for k, d in agg.items():
        print("   ", k, {c: round(v / len(n[k])) for c, v in d.items()}, "dispatches", len(n[k]))
PY
done
