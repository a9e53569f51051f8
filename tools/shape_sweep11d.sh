cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 300 python bench.py "$@" --no-c5 --steps 3 --warmup 1 --no-cpu-baseline --no-scopes --no-single-call > gpurun_out/s_$tag.json 2> gpurun_out/s_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s_$tag.json").read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    top=sorted(k.items(), key=lambda kv:-kv[1])[:5]
    print("$tag", d["ms_per_step"], top, "mism", d["parity"]["statistic_mismatches"], d["parity"]["p_value_max_rel_err"])
except Exception as e:
    print("$tag", "FAILED", e, open("gpurun_out/s_$tag.err").read()[-300:])
PY
}
K="--cells 1000000 --genes 2400 --groups 10"
run clus_dense_cont_ovo $K --workload c2 --values continuous --sparsity 0.9
run clus_csr_cont_ovo $K --workload c3 --format csr --values continuous
run clus_csc_cont_ovo $K --workload c3 --values continuous
run clus_dense_cont_ovo_s50 $K --workload c2 --values continuous
