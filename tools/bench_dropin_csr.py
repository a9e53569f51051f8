#!/usr/bin/env python
"""Drop-in call with a HOST-resident CSR matrix (the usual AnnData.X): where does the wall-clock go?"""
import argparse, json, sys, time
from pathlib import Path
import numpy as np, pandas as pd
from scipy import sparse
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=300_000); ap.add_argument("--genes", type=int, default=8_000)
ap.add_argument("--groups", type=int, default=2_000); ap.add_argument("--test", default="ovo")
ap.add_argument("--values", default="continuous")
a = ap.parse_args()
import torch
from bench import make_labels, make_matrix
from illico_amd import AnnDataLite, asymptotic_wilcoxon
from illico_amd.utils.ranking import check_indices_sorted_per_parcel
N, M, G = a.cells, a.genes, a.groups
codes = make_labels(N, G, 0)
labels = pd.Categorical(np.where(codes == 0, "non-targeting", np.char.add("pert_", np.char.zfill(codes.astype(str), 5))))
blocks = []
for j in range(0, M, 1000):
    blocks.append(sparse.csr_matrix(make_matrix(torch, N, min(1000, M - j), 0.9, j, torch.device("cuda", 0), continuous=a.values == "continuous").cpu().numpy()))
X = sparse.hstack(blocks, format="csr"); X.sort_indices()
adata = AnnDataLite(X, obs=pd.DataFrame({"pert": labels}))
ref = "non-targeting" if a.test == "ovo" else None
log1p = a.values == "continuous"
asymptotic_wilcoxon(adata, log1p, "pert", ref)  # warm-up
t = {}
t0 = time.perf_counter(); check_indices_sorted_per_parcel(X.indices, X.indptr); t["sorted_check_host_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); df = asymptotic_wilcoxon(adata, log1p, "pert", ref); t["drop_in_call_s"] = time.perf_counter() - t0
from illico_amd._lib import get_engine
from illico_amd.utils.groups import encode_and_count_groups
eng = get_engine(); _, g = encode_and_count_groups(adata.obs["pert"], ref); eng.set_groups(g)
t0 = time.perf_counter(); eng.run_sparse("csr", X.data, X.indices, X.indptr, X.shape, 0, M, is_log1p=log1p); t["engine_with_transfers_s"] = time.perf_counter() - t0
t["nnz"] = int(X.nnz); t["input_GB"] = (X.data.nbytes + X.indices.nbytes + X.indptr.nbytes) / 1e9; t["rows_in_result"] = len(df)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}))
