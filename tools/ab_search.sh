#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# A/B helper (dev tool): rebuild with extra -D flags for ia_snarf.hip and time the compact search alone
for flags in "$@"; do
  ab_rebuild ia_search.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  timeout 120 python tools/bench_search.py "$flags" 16 2>&1 | tail -1
  timeout 120 python tools/bench_search.py "$flags" 4 2>&1 | tail -1
done
