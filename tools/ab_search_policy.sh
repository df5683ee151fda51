#!/bin/bash
# round 6: k_search with default-policy vs `nt` (L1-bypassing) record fetches, per launch position of the
# frame (probe launch: L1 hit 72 %; render launches: 92 %), from the kernel trace of an eager one-frame-in-flight bench run.
source "$(dirname "$0")/ab_lib.sh"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for flags in "" "-DIA_SEARCH_POL=1"; do
  cd $R; ab_rebuild ia_search.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp
  env $(ab_flags_env ia_search.hip "$flags") timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o r -- python $R/bench.py --steps 20 --warmup 4 --spinup-max-ms 50 --cpu-frames 0 --train-steps 0 --no-graph --in-flight 1 --no-profile > /tmp/pp.log 2>&1
  echo "=== ia_search.hip [$flags]"
  python $R/tools/search_launches.py $(find /tmp/pp -name "r_kernel_trace.csv" | head -1) 6
done
cd $R; ab_rebuild ia_search.hip ""
