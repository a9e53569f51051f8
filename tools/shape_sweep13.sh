# Round 4: fully dense continuous values (scaled data: no zeros at all) at C2 and C5-shard shape
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
run c2_full_ovo --workload c2 --values continuous --sparsity 0.0
run c2_full_ovr --workload c2 --values continuous --sparsity 0.0 --test ovr
run c5s_half_ovo --workload c5shard --values continuous
run c5s_full_ovo --workload c5shard --values continuous --sparsity 0.0
run c5s_full_ovr --workload c5shard --values continuous --sparsity 0.0 --test ovr
run c5s_s90_ovo --workload c5shard --values continuous --sparsity 0.9
