#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# dev tool: rebuild <file> with flags, print the probe / render split (tools/probe_stats.py) and a short frame bench
f=$1; shift
for flags in "$@"; do
  ab_rebuild $f "$flags" || { echo "build failed: [$flags]"; continue; }
  echo "=== [$flags]"; timeout 100 python tools/probe_stats.py 2>&1 | tail -2 | cut -c1-75
  timeout 100 python bench.py --steps 100 --warmup 10 --cpu-frames 0 --train-steps 0 --no-profile 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'fps', round(d['ms_per_step'],3))"
done
