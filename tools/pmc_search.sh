#!/bin/bash
# dev tool: PMC passes over a short bench run, summarised for k_search / k_field / k_encode_xcd
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_s$i -o r -- python $R/bench.py --steps 3 --warmup 2 --cpu-frames 0 --train-steps 0 --no-graph > $R/gpurun_out/pmc_s$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_s*/**/*counter_collection.csv", recursive=True)):
    tot=collections.defaultdict(float); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        for nm in ("k_search","k_field","k_encode_xcd","k_march_compact"):
            if nm in k:
                tot[(nm,r["Counter_Name"])]+=float(r["Counter_Value"]); n[(nm,r["Counter_Name"])].add(r["Dispatch_Id"])
    for k,v in sorted(tot.items()): print(k, "per launch %.4g (%d launches)"%(v/len(n[k]), len(n[k])))
PY
