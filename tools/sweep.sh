#!/bin/bash
# usage: tools/sweep.sh <option-name> v1 v2 ...   (runs bench.py with --engine-option name=v and prints kernel times)
opt=$1; shift
for g in "$@"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --engine-option $opt=$g 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$opt=$g', d['ms_per_step'], d['roofline']['all_kernels_ms_per_step'])"
done
