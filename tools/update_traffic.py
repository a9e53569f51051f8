#!/usr/bin/env python3
"""Put the PMC traffic of one profiled workload into profiles/traffic.json (what bench.py copies into roofline.traffic).

    python tools/update_traffic.py profiles/r03_c3 k_csc_counts "k_csc_counts<float, int, false" --workload c3 [bench.py workload flags]

Reads <dir>/summary.json (tools/profile_bench.sh + tools/summarize_prof.py), takes the kernel whose name contains the given
substring, and replaces the entry with the same kernel_id and workload key."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

prof_dir, kernel_id, needle = sys.argv[1], sys.argv[2], sys.argv[3]
args = bench.parse(sys.argv[4:])
summary = json.loads((ROOT / prof_dir / "summary.json").read_text())
names = [k for k in summary["kernels"] if needle in k and "hbm_bytes_per_launch" in summary["kernels"][k]]
if len(names) != 1:
    raise SystemExit(f"{len(names)} kernels of {prof_dir} match {needle!r}: {names}")
e = summary["kernels"][names[0]]
wl = {"workload": args.workload, "cells": args.cells, "genes_per_gpu": args.genes, "groups": args.groups, "test": args.test, "format": args.fmt,
      "values": args.values, "sparsity": args.sparsity}
if args.mean_max != 15.0:
    wl["mean_max"] = args.mean_max
entry = {"kernel_id": kernel_id, "kernel": names[0], "workload": wl, "launches_per_step": 1,
         "hbm_bytes_per_launch": int(e["hbm_bytes_per_launch"]), "hbm_read_bytes_corrected": int(e["hbm_read_bytes_corrected"]),
         "hbm_write_bytes": int(e["hbm_write_bytes"]), "avg_ms_profiled": round(e.get("avg_ms", 0.0), 4),
         "source": f"{prof_dir}/summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x 2 on gfx950 as "
                   "MI355X_MICROARCH.md prescribes), one launch per step"}
tf = ROOT / "profiles" / "traffic.json"
doc = json.loads(tf.read_text())
doc["entries"] = [x for x in doc["entries"] if not (x.get("kernel_id") == kernel_id and x.get("workload") == wl)] + [entry]
tf.write_text(json.dumps(doc, indent=1) + "\n")
print(json.dumps(entry, indent=1))
