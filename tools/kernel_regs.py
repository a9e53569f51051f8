#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels in the built library's translation units (illico_amd/csrc/_build/*.o, code-object metadata):
   python tools/kernel_regs.py [substring ...]"""
import re, subprocess, sys, tempfile
from pathlib import Path
L = "/opt/rocm/lib/llvm/bin/"
objs = sorted((Path(__file__).resolve().parent.parent / "illico_amd" / "csrc" / "_build").glob("*.o"))  # one code object per translation unit
notes = ""
with tempfile.TemporaryDirectory() as d:
    for o in objs:
        subprocess.run([L + "llvm-objcopy", f"--dump-section=.hip_fatbin={d}/fat.bin", str(o), f"{d}/copy.o"], check=True, capture_output=True)  # (no output name: objcopy rewrites its input, and the build sees a fresh object)
        tg = [t for t in subprocess.run([L + "clang-offload-bundler", "--list", "--type=o", f"--input={d}/fat.bin"], capture_output=True, text=True).stdout.split() if "gfx950" in t][0]
        subprocess.run([L + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={d}/fat.bin", f"--targets={tg}", f"--output={d}/k.co"], check=True)
        notes += subprocess.run([L + "llvm-readelf", "--notes", f"{d}/k.co"], capture_output=True, text=True).stdout
want = sys.argv[1:]
for e in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", e).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    if want and not any(w in dem for w in want):
        continue
    g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, e) or [0, 0])[1])
    agpr = int(re.match(r"\s*(\d+)", e).group(1))
    print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):3d} agpr {agpr:3d} sgpr {g('sgpr_count'):3d} "
          f"scratch {g('private_segment_fixed_size'):4d} spilled {g('vgpr_spill_count'):3d} lds {g('group_segment_fixed_size'):6d}")
