#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# dev tool: rebuild ia_snarf.hip with flags and print the probe / render split of k_search
for flags in "$@"; do
  ab_rebuild ia_search.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  echo "=== [$flags]"; timeout 100 python tools/probe_stats.py 2>&1 | tail -2 | cut -c1-75
done
