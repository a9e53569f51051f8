# Round 4: extreme aspect ratios and group counts that no earlier sweep ran (tall: 2M cells x 1200 genes; wide: 20k cells x 120k genes;
# 30 000 groups of ten cells).   bash tools/shape_sweep10.sh > gpurun_out/sweep10.txt
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 300 python bench.py "$@" --no-c5 --steps 3 --warmup 1 --no-cpu-baseline --no-scopes --no-single-call > gpurun_out/s_$tag.json 2> gpurun_out/s_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s_$tag.json").read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    top=sorted(k.items(), key=lambda kv:-kv[1])[:5]
    print("$tag", d["ms_per_step"], "GB", round(d["roofline"].get("algorithmic_bytes_per_launch", 0)/1e9, 2) if d["roofline"].get("algorithmic_bytes_per_launch") else "", top, "mism", d["parity"]["statistic_mismatches"], d["parity"]["p_value_max_rel_err"])
except Exception as e:
    print("$tag", "FAILED", e, open("gpurun_out/s_$tag.err").read()[-300:])
PY
}
T="--cells 2000000 --genes 1200 --groups 2000"
run tall_dense $T --workload c2
run tall_dense_ovr $T --workload c4
run tall_dense_cont_ovr $T --workload c2 --values continuous --test ovr
run tall_dense_cont $T --workload c2 --values continuous
run tall_csr $T --workload c3 --format csr
run tall_csr_ovr $T --workload c3 --format csr --test ovr
run tall_csc $T --workload c3
run tall_csr_cont $T --workload c3 --format csr --values continuous
run tall_csc_cont_ovr $T --workload c3 --values continuous --test ovr
W="--cells 20000 --genes 120000 --groups 100"
run wide_dense $W --workload c2
run wide_dense_ovr $W --workload c4
run wide_dense_cont $W --workload c2 --values continuous
run wide_csr $W --workload c3 --format csr
run wide_csc_ovr $W --workload c3 --test ovr
run wide_csr_cont_ovr $W --workload c3 --format csr --values continuous --test ovr
S="--groups 30000"
run tiny_groups_dense $S --workload c2
run tiny_groups_dense_ovr $S --workload c4
run tiny_groups_csc $S --workload c3
run tiny_groups_csr $S --workload c3 --format csr
run tiny_groups_dense_cont $S --workload c2 --values continuous
