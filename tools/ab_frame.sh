#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# A/B helper (dev tool): rebuild ONE source file with extra -D flags and time the render-only bench
#   tools/ab_frame.sh ia_field.hip "-DIA_ENC_S=2" "-DIA_ENC_S=1"
f=$1; shift
for flags in "" "$@"; do
  ab_rebuild $f "$flags" || { echo "build failed: [$flags]"; continue; }
  timeout 200 python bench.py --train-steps 0 --cpu-frames 0 --no-profile --steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flags=[$flags]', round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms; 1-in-flight', round(d['one_frame_in_flight']['frame_latency_ms'],3), 'ms')"
done
