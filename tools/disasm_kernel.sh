#!/bin/bash
# Usage: tools/disasm_kernel.sh <unit, e.g. core> <kernel-name regex>   -- prints the gfx950 ISA of the matching kernels of one built unit
set -e
L=/opt/rocm/lib/llvm/bin
d=$(mktemp -d)
$L/llvm-objcopy --dump-section=.hip_fatbin=$d/fat.bin illico_amd/csrc/_build/$1.o $d/copy.o
tg=$($L/clang-offload-bundler --list --type=o --input=$d/fat.bin | grep gfx950)
$L/clang-offload-bundler --unbundle --type=o --input=$d/fat.bin --targets=$tg --output=$d/k.co
$L/llvm-objdump -d --no-show-raw-insn $d/k.co | awk -v pat="$2" '/^[0-9a-f]+ <.*>:$/ { on = ($0 ~ pat) } on { print }'
rm -rf $d
