# Round 4, last sweep: CSR shapes around C3 (the container AnnData holds by default) that no earlier sweep ran at full size.
# Usage (GPU box): bash tools/shape_sweep8.sh > gpurun_out/sweep8.txt
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
run csr_s99 --workload c3 --format csr --sparsity 0.99
run csr_s99_ovr --workload c3 --format csr --sparsity 0.99 --test ovr
run csr_s70 --workload c3 --format csr --sparsity 0.7
run csr_s70_ovr --workload c3 --format csr --sparsity 0.7 --test ovr
run csr_g50 --workload c3 --format csr --groups 50
run csr_g50_ovr --workload c3 --format csr --groups 50 --test ovr
run csr_g10000 --workload c3 --format csr --groups 10000
run csr_g10000_ovr --workload c3 --format csr --groups 10000 --test ovr
run csr_cont_s99 --workload c3 --format csr --values continuous --sparsity 0.99
run csr_cont_s70_ovr --workload c3 --format csr --values continuous --sparsity 0.7 --test ovr
run csr_cont_g50 --workload c3 --format csr --values continuous --groups 50
run csr_cont_g10000_ovr --workload c3 --format csr --values continuous --groups 10000 --test ovr
run csr_nb --workload c3 --format csr --values nb
run csr_nb_ovr --workload c3 --format csr --values nb --test ovr
run csr_mean40 --workload c3 --format csr --mean-max 40
run csc_s99 --workload c3 --sparsity 0.99
