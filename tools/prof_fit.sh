#!/bin/bash
# dev tool: rocprofv3 kernel stats of the fit stage (SMPLDeformer + SMPLParamEmbedding), GPU time per step against wall time
cd /tmp; export TMPDIR=/tmp; export PYTHONPATH=$GRAFT_REPO_ROOT; rm -rf /tmp/pf
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o r -- python -m instantavatar_amd.drivers.fit --synthetic --steps ${1:-100} --res 256 --out /tmp/fitp 2>&1 | grep "it/s" | tail -2
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/pf/r_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
n=${1:-100}
print("total GPU kernel time %.1f ms over %d steps (+setup) -> %.2f ms/step; launches per step %.0f" % (tot/1e6, n, tot/1e6/n, sum(int(r["Calls"]) for r in rows)/n))
for r in rows[:18]:
    print("%-110s calls %6s avg %9.1f us total %8.1f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
