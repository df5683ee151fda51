#!/bin/bash
# round 6: k_precompute with the channel-last transform records staged through LDS (every store instruction
# contiguous) and with more weight planes in flight, in isolation (tools/bench_precompute.py: events on the launch stream).
source "$(dirname "$0")/ab_lib.sh"
R=$GRAFT_REPO_ROOT
for flags in "-DIA_PRE_LDS_STORE=0" "" "-DIA_PRE_UNROLL=8"; do
  cd $R; ab_rebuild ia_snarf.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  env $(ab_flags_env ia_snarf.hip "$flags") timeout 120 python tools/bench_precompute.py "[$flags]" 2>&1 | grep "tag="
done
cd $R; ab_rebuild ia_snarf.hip ""
