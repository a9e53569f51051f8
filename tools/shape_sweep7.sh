cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 300 python bench.py "$@" --no-c5 --steps 3 --warmup 1 --no-cpu-baseline --no-scopes --no-single-call > gpurun_out/s_$tag.json 2> gpurun_out/s_$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/s_$tag.json")); k=d["roofline"]["all_kernels_ms_per_step"]
    top=sorted(k.items(), key=lambda kv:-kv[1])[:6]
    print("$tag", d["ms_per_step"], top, "mism", d["parity"]["statistic_mismatches"], d["parity"]["p_value_max_rel_err"])
except Exception as e:
    print("$tag", "FAILED", e, open("gpurun_out/s_$tag.err").read()[-300:])
PY
}
run cont_csc_g300 --workload c3 --values continuous --groups 300
run cont_csc_g50 --workload c3 --values continuous --groups 50
run cont_csc_ovr_g300 --workload c3 --values continuous --groups 300 --test ovr
run cont_csr_g300 --workload c3 --format csr --values continuous --groups 300
