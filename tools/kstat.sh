#!/bin/bash
# dev tool: per-kernel average durations of a short eager bench run (rocprofv3 --stats)
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-frames 0 --train-steps 0 --no-graph --no-profile >/dev/null 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/pp/r_kernel_stats.csv')):
    n=r['Name'].split('(')[0]
    if n.startswith('k_') or 'k_' in n[:12]: print(f"{n[:44]:44s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
