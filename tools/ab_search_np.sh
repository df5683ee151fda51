#!/bin/bash
# round 6: does a smaller workgroup tile help the SMALL k_search launches of the frame (iterations 0, 4, 5 ...)?
# per-position table for IA_SEARCH_NP = 64 (default) / 32 / 16 from an eager one-frame-in-flight run, plus the two-in-flight rate.
source "$(dirname "$0")/ab_lib.sh"
R=$GRAFT_REPO_ROOT
for flags in "" "-DIA_SEARCH_NP=32" "-DIA_SEARCH_NP=16"; do
  cd $R; ab_rebuild ia_search.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp
  env $(ab_flags_env ia_search.hip "$flags") timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -o r -- python $R/bench.py --steps 20 --warmup 4 --spinup-max-ms 50 --cpu-frames 0 --train-steps 0 --no-graph --in-flight 1 --no-profile > /tmp/pp.log 2>&1
  echo "=== ia_search.hip [$flags]"
  python $R/tools/search_launches.py $(find /tmp/pp -name "r_kernel_trace.csv" | head -1) 6
  cd $R; env $(ab_flags_env ia_search.hip "$flags") timeout 200 python bench.py --steps 100 --warmup 10 --cpu-frames 0 --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two in flight, graph replay: %.1f frames/s (%.3f ms per frame)' % (d['value'], d['ms_per_step']))"
done
cd $R; ab_rebuild ia_search.hip ""
