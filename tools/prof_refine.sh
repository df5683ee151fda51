#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_refine
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refine -o r -- python $R/tools/prof_refine.py 60 $1 > $O/prof_refine.log 2>&1
grep it_per_sec $O/prof_refine.log
python - <<'PY'
import csv, os
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(R+"/gpurun_out/prof_refine/r_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
ours=[r for r in rows if r["Name"].startswith("void k_") or r["Name"].startswith("k_")]
print("total %.1f ms; ia kernels %.1f ms in %d calls; other %.1f ms in %d calls" % (tot/1e6, sum(float(r["TotalDurationNs"]) for r in ours)/1e6, sum(int(r["Calls"]) for r in ours),
      (tot-sum(float(r["TotalDurationNs"]) for r in ours))/1e6, sum(int(r["Calls"]) for r in rows)-sum(int(r["Calls"]) for r in ours)))
for r in rows[:34]:
    print("%-70s calls %6s total %8.2f ms avg %8.1f us" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
