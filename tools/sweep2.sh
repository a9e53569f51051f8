#!/bin/bash
# usage: tools/sweep2.sh "<bench args>" <option-name> v1 v2 ...
args=$1; opt=$2; shift; shift
for g in "$@"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline $args --engine-option $opt=$g 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$opt=$g', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
done
