#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# A/B helper (dev tool): rebuild with extra -D flags for ia_snarf.hip and time ia_precompute alone
for flags in "$@"; do
  ab_rebuild ia_snarf.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  timeout 120 python tools/bench_precompute.py "$flags" 2>&1 | tail -1
done
