#!/bin/bash
# Usage (GPU box): tools/micro/pmc_ocb.sh [harness args]  -- SQ counters of the packed-route kernels in the microbenchmark
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ocb
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/a -- tools/micro/ovo_compact_bench "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- tools/micro/ovo_compact_bench "$@" > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(sys.argv[1] + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "compact" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()): print("   %-24s %.4g" % (c, v / max(1, len(n[k][c]))))
PY
