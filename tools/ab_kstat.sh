#!/bin/bash
source "$(dirname "$0")/ab_lib.sh"
# dev tool: rebuild one source with flags, print the average duration of the kernels matching $2
f=$1; pat=$2; shift 2
for flags in "$@"; do
  ab_rebuild $f "$flags" || { echo "build failed: [$flags]"; continue; }
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --spinup-max-ms 100 --cpu-frames 0 --train-steps 0 --no-graph --no-profile >/dev/null 2>&1
  echo "=== [$flags]"; grep -E "$pat" /tmp/pp/r_kernel_stats.csv | awk -F, '{printf "%s calls %s avg %.1f us\n", substr($1,1,40), $(NF-6), $(NF-4)/1000}'
  cd $GRAFT_REPO_ROOT
done
