#!/bin/bash
# per-kernel time of the inference frame: rocprofv3 --kernel-trace --stats over a render-only bench run, one frame in flight
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_frame
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_frame -o r -- python $R/bench.py --steps 100 --warmup 10 --cpu-frames 0 --train-steps 0 --in-flight 1 --no-profile > $O/prof_frame.log 2>&1
tail -1 $O/prof_frame.log | cut -c1-300
python - <<'PY'
import csv, os
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(R+"/gpurun_out/prof_frame/r_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot/1e6))
for r in rows[:40]:
    print("%-60s calls %6s total %8.2f ms avg %8.1f us  %5.1f%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
