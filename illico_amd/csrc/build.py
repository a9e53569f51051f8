"""Build libillico_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
SO = HERE / "libillico_hip.so"
SOURCES = ["illico_hip.hip"]
HEADERS = ["common.h", "kernels_ovo.h", "kernels_ovo_compact.h", "kernels_ovo_counts.h", "kernels_ovo_fused.h", "kernels_ovr.h", "kernels_finalize.h", "kernels_sparse.h", "kernels_csc_gene.h", "kernels_csc_counts.h", "kernels_csc_ovr.h", "kernels_ovr_parts.h", "kernels_sums.h", "kernels_leftover.h", "ovr_driver.h",
           "sparse_driver.h", "../../include/illico_hip.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"]


STAMP = HERE / "libillico_hip.stamp"  # the flags the library on disk was built with (a development build must not pass for a full one)


def _flags() -> list[str]:
    flags = list(FLAGS)
    if os.environ.get("ILLICO_DEV_F32_ONLY") == "1":  # development: float32 / int32-index kernels only, ~4x faster to compile
        flags.append("-DILLICO_DEV_F32_ONLY")
    return flags


def needs_build() -> bool:
    if not SO.exists() or not STAMP.exists() or STAMP.read_text() != " ".join(_flags()):
        return True
    t = SO.stat().st_mtime
    return any((HERE / f).stat().st_mtime > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = _flags()
    cmd = [hipcc, *flags, "-o", str(SO), *[str(HERE / s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
    STAMP.write_text(" ".join(flags))
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
