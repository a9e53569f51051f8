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
T="--cells 2000000 --genes 1200 --groups 2000"
run tall_dense_cont $T --workload c2 --values continuous
run tall_dense_cont_s90 $T --workload c2 --values continuous --sparsity 0.9
run tall_csr_cont $T --workload c3 --format csr --values continuous
run tall_csc_cont $T --workload c3 --values continuous
