#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# A/B helper (dev tool): rebuild libinstantavatar_hip.so on the GPU box with extra -D flags for one
# source file and run the frame bench.   usage: tools/ab_build.sh <file.hip> "<flags A>" "<flags B>" ...
f=$1; shift
for flags in "$@"; do
  ab_rebuild $f "$flags" || { echo "build failed: [$flags]"; continue; }
  for i in 1 2; do timeout 120 python bench.py --steps 40 --warmup 5 --cpu-frames 0 --train-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); o=d['roofline']['other']; print('[$flags]', round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms  k_search', round(o['k_search']['avg_launch_us'],1), 'us  k_field', round(o['k_field']['avg_launch_us'],1), 'us')"; done
done
