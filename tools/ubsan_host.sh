#!/bin/bash
# Host-side undefined-behaviour check of the drivers (core.hip, dense_driver.h, sparse_driver.h, keyed_impl.h: pointer arithmetic on scratch,
# 64-bit size products, shifts).  The HOST half of the library is built with UBSan in trap mode (no runtime library needed in the
# python process; the device code is compiled as usual: GPU sanitizers are not available on this pool), swapped in for the product
# library on the GPU box's scratch copy, and the whole GPU suite runs against it: an undefined operation ends the run with SIGILL
# at the offending test.
#   here:        tools/ubsan_host.sh build            -> tools/micro/libillico_ubsan.so (git-ignored, travels with gpurun; ~6 min)
#   on the box:  tools/ubsan_host.sh run [pytest args] -> gpurun_out/ubsan_tests.log
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
case "${1:-}" in
build)
    # every translation unit of the library (illico_amd/csrc/build.py: UNITS) in one hipcc call
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -fPIC -shared -std=c++17 -ffp-contract=off -Wno-unused-value \
        -fsanitize=signed-integer-overflow,shift,bounds,alignment,null,pointer-overflow,integer-divide-by-zero,float-cast-overflow \
        -fsanitize-trap=all -fno-gpu-sanitize -Wl,--version-script="$R/illico_amd/csrc/exports.map" \
        -o "$R/tools/micro/libillico_ubsan.so" "$R"/illico_amd/csrc/*.hip
    ;;
run)
    shift
    cp "$R/tools/micro/libillico_ubsan.so" "$R/illico_amd/csrc/libillico_hip.so"   # the box's copy is scratch
    mkdir -p "$R/gpurun_out"
    cd "$R" && python -m pytest tests -v -m gpu -p no:cacheprovider "$@" > gpurun_out/ubsan_tests.log 2>&1
    tail -3 gpurun_out/ubsan_tests.log
    ;;
*) echo "usage: $0 build | run [pytest args]"; exit 2 ;;
esac
