# Round 4: everyday sizes -- a scanpy data set (100 000 cells x 30 000 genes, 30 clusters), a small Perturb-seq screen (40 000 x 20 000, 300
# perturbations), a toy (500 x 20 000, 5 groups)
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; timeout -k 10 300 python bench.py "$@" --no-c5 --steps 3 --warmup 1 --no-cpu-baseline --no-scopes --no-single-call > gpurun_out/s_$tag.json 2> gpurun_out/s_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s_$tag.json").read().strip().splitlines()[-1]); k=d["roofline"]["all_kernels_ms_per_step"]
    top=sorted(k.items(), key=lambda kv:-kv[1])[:5]
    print("$tag", d["ms_per_step"], "first", d["timing_scopes"].get("first_call_ms") if d.get("timing_scopes") else "", top, "mism", d["parity"]["statistic_mismatches"], d["parity"]["p_value_max_rel_err"])
except Exception as e:
    print("$tag", "FAILED", e, open("gpurun_out/s_$tag.err").read()[-300:])
PY
}
S="--cells 100000 --genes 30000 --groups 30"
run scanpy_csr_cont_ovr $S --workload c3 --format csr --values continuous --test ovr --sparsity 0.93
run scanpy_csr_counts_ovr $S --workload c3 --format csr --test ovr --sparsity 0.93
run scanpy_csr_nb_ovr $S --workload c3 --format csr --values nb --test ovr --sparsity 0.93
run scanpy_csc_cont_ovr $S --workload c3 --values continuous --test ovr --sparsity 0.93
run scanpy_dense_cont_ovr $S --workload c2 --values continuous --test ovr --sparsity 0.93
run scanpy_csr_cont_ovo $S --workload c3 --format csr --values continuous --sparsity 0.93
P="--cells 40000 --genes 20000 --groups 300"
run screen_csr_counts_ovo $P --workload c3 --format csr
run screen_csr_cont_ovo $P --workload c3 --format csr --values continuous
run screen_dense_cont_ovo $P --workload c2 --values continuous --sparsity 0.9
T="--cells 500 --genes 20000 --groups 5"
run toy_csr_cont_ovr $T --workload c3 --format csr --values continuous --test ovr
run toy_dense_counts_ovo $T --workload c2
