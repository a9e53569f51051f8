"""Build libillico_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

The library is thirteen translation units -- the context / C-ABI (core.hip), the launchers that depend on the key type only
(keyed_u32 / keyed_u64), and one unit per value type for the dense and for the sparse drivers -- compiled in parallel into
_build/*.o and linked.  A unit is recompiled when it, or a header its last compile read (the -MD dependency file), changed:
an edit to one kernel family rebuilds the units that include it, side by side, in about a minute instead of five.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
SO = HERE / "libillico_hip.so"
OBJ = HERE / "_build"
UNITS = ["core", "keyed_u32", "keyed_u64", "keyed_coop", "dense_f32", "dense_f64", "dense_i32", "dense_i64", "dense_u8",
         "sparse_f32", "sparse_f64", "sparse_i32", "sparse_i64"]
DEV_UNITS = ["core", "keyed_u32", "keyed_coop", "dense_f32", "dense_u8", "sparse_f32"]  # ILLICO_DEV_F32_ONLY=1: float32 values, int32 indices
CFLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"]
LDFLAGS = ["--offload-arch=gfx950", "-fPIC", "-shared", "-Wl,-z,defs", f"-Wl,--version-script={HERE / 'exports.map'}"]

STAMP = HERE / "libillico_hip.stamp"  # the flags the library on disk was built with (a development build must not pass for a full one)


def _dev() -> bool:
    return os.environ.get("ILLICO_DEV_F32_ONLY") == "1"


def _flags() -> list[str]:
    flags = list(CFLAGS)
    if _dev():  # development: float32 / int32-index kernels only
        flags.append("-DILLICO_DEV_F32_ONLY")
    flags += os.environ.get("ILLICO_EXTRA_CFLAGS", "").split()  # kernel experiments (-DTRG_WIN=16 ...): the stamp records them
    return flags


def _units() -> list[str]:
    return DEV_UNITS if _dev() else UNITS


def _deps(unit: str) -> list[Path]:
    d = OBJ / f"{unit}.d"
    if not d.exists():
        return []
    words = d.read_text().replace("\\\n", " ").split()
    return [Path(w) for w in words[1:] if not w.endswith(":")]


def _unit_stale(unit: str, tag: str) -> bool:
    obj, tagf = OBJ / f"{unit}.o", OBJ / f"{unit}.flags"
    if not obj.exists() or not tagf.exists() or tagf.read_text() != tag:
        return True
    deps = _deps(unit)
    if not deps:
        return True
    t = obj.stat().st_mtime
    return any((not p.exists()) or p.stat().st_mtime > t for p in deps)


def needs_build() -> bool:
    tag = " ".join(_flags())
    if not SO.exists() or not STAMP.exists() or STAMP.read_text() != tag + " | " + " ".join(_units()):
        return True
    if any(_unit_stale(u, tag) for u in _units()):
        return True
    # objects newer than the library: every unit compiled but the link failed or was interrupted -- the library on disk is the OLD one
    t = SO.stat().st_mtime
    return any((OBJ / f"{u}.o").stat().st_mtime > t for u in _units())


def _compile(unit: str, hipcc: str, flags: list[str], verbose: bool) -> tuple[str, int, str]:
    cmd = [hipcc, *flags, "-c", "-MD", "-MF", str(OBJ / f"{unit}.d"), "-o", str(OBJ / f"{unit}.o"), str(HERE / f"{unit}.hip")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode == 0:
        (OBJ / f"{unit}.flags").write_text(" ".join(flags))
    return unit, r.returncode, r.stdout + r.stderr


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> Path:
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags, units = _flags(), _units()
    tag = " ".join(flags)
    OBJ.mkdir(exist_ok=True)
    todo = [u for u in units if force or _unit_stale(u, tag)]
    jobs = jobs or int(os.environ.get("ILLICO_BUILD_JOBS", "0")) or min(len(todo) or 1, os.cpu_count() or 4)
    # the long units first: the pool then ends on the short ones
    order = {"sparse": 0, "dense_": 1, "keyed": 2, "core": 3}
    todo.sort(key=lambda u: next(v for k, v in order.items() if u.startswith(k)))
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        results = list(pool.map(lambda u: _compile(u, hipcc, flags, verbose), todo))
    bad = [(u, out) for u, rc, out in results if rc != 0]
    if bad:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {u} ---\n{out}" for u, out in bad))
    STAMP.unlink(missing_ok=True)  # (written again only after a successful link)
    cmd = [hipcc, *LDFLAGS, "-o", str(SO), *[str(OBJ / f"{u}.o") for u in units]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    STAMP.write_text(tag + " | " + " ".join(units))
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
