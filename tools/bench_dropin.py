#!/usr/bin/env python
"""Drop-in call timing with HOST-resident input (what a user of illico.asymptotic_wilcoxon passes):
group encoding, H2D + engine, DataFrame assembly (SURVEY.md 8d timing scopes ii and iii)."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np, pandas as pd
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=300_000); ap.add_argument("--genes", type=int, default=2_000)
ap.add_argument("--groups", type=int, default=2_000); ap.add_argument("--test", default="ovo")
a = ap.parse_args()
import torch
from bench import make_labels, make_matrix
from illico_amd import AnnDataLite, asymptotic_wilcoxon
from illico_amd.utils.groups import encode_and_count_groups
N, M, G = a.cells, a.genes, a.groups
codes = make_labels(N, G, 0)
labels = np.where(codes == 0, "non-targeting", np.char.add("pert_", np.char.zfill(codes.astype(str), 5)))
X = make_matrix(torch, N, M, 0.5, 0, torch.device("cuda", 0)).cpu().numpy()
adata = AnnDataLite(X, obs=pd.DataFrame({"pert": labels}))
ref = "non-targeting" if a.test == "ovo" else None
asymptotic_wilcoxon(adata, False, "pert", ref)  # warm-up (context, scratch)
t = {}
t0 = time.perf_counter(); encode_and_count_groups(labels, ref); t["encode_groups_s"] = time.perf_counter() - t0
t0 = time.perf_counter(); df = asymptotic_wilcoxon(adata, False, "pert", ref); t["drop_in_call_s"] = time.perf_counter() - t0
from illico_amd._lib import get_engine
eng = get_engine(); _, g = encode_and_count_groups(labels, ref); eng.set_groups(g)
t0 = time.perf_counter(); eng.run_dense(X, 0, M); t["engine_with_transfers_s"] = time.perf_counter() - t0
Xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
t0 = time.perf_counter(); eng.run_dense(Xd, 0, M, device_out=True); eng.synchronize(); t["engine_device_resident_s"] = time.perf_counter() - t0
t["input_GB"] = X.nbytes / 1e9; t["rows_in_result"] = len(df)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}))
