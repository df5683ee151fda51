#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# A/B helper (dev tool): rebuild one source with extra -D flags and run the train-only bench
f=$1; shift
for flags in "$@"; do
  ab_rebuild $f "$flags" || { echo "build failed: [$flags]"; continue; }
  for i in 1 2; do timeout 200 python bench.py --train-only --steps 300 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d['train']; print('[$flags]', round(t['it_per_sec'],1), 'it/s', t.get('launch_mode'), t['mse_last'])"; done
done
