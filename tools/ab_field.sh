#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# A/B helper (dev tool): rebuild ia_field.hip with extra -D flags on the GPU box, run tools/bench_field.py
for flags in "$@"; do
  ab_rebuild ia_field.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  echo "=== [$flags]"
  python tools/bench_field.py 2>&1 | grep -E "uniform.*(65536|262144|1048576)"
done
