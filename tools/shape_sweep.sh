cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 200 python bench.py "$@" --no-c5 --steps 5 --warmup 2 --no-cpu-baseline --no-scopes > gpurun_out/s_$tag.json 2> gpurun_out/s_$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/s_$tag.json")); k=d["roofline"]["all_kernels_ms_per_step"]
    top=sorted(k.items(), key=lambda kv:-kv[1])[:3]
    print("$tag", d["ms_per_step"], "pipe", d["roofline"]["pipeline_frac"], top, "mism", d["parity"]["statistic_mismatches"])
except Exception as e:
    print("$tag", "FAILED", e, open("gpurun_out/s_$tag.err").read()[-300:])
PY
}
run dense_g50 --workload c2 --groups 50
run dense_g10000 --workload c2 --groups 10000
run dense_ovr_g50 --workload c4 --groups 50
run dense_ovr_g10000 --workload c4 --groups 10000
run csc_g50 --workload c3 --groups 50
run csc_g5000 --workload c3 --groups 5000
run csc_g10000 --workload c3 --groups 10000
run csc_ovr_g50 --workload c3 --test ovr --groups 50
run csc_ovr_g5000 --workload c3 --test ovr --groups 5000
run csr_g300 --workload c3 --format csr --groups 300
run csr_g5000 --workload c3 --format csr --groups 5000
run csr_ovr_g300 --workload c3 --format csr --test ovr --groups 300
run dense_small_n --workload c2 --cells 20000 --genes 30000 --groups 100
run csc_small_n --workload c3 --cells 20000 --genes 30000 --groups 100
run csc_dense50 --workload c3 --sparsity 0.5
run csc_sparse99 --workload c3 --sparsity 0.99
