#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_enc_$n -o r -- python $R/tools/pmc_encode.py > $R/gpurun_out/pmc_enc_$n.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_enc_*/**/*counter_collection.csv", recursive=True)):
    tot=collections.defaultdict(float); n=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_hashgrid" in k or "k_encode" in k:
            tot[(k[:40],r["Counter_Name"])]+=float(r["Counter_Value"]); n[(k[:40],r["Counter_Name"])].add(r["Dispatch_Id"])
    for k,v in sorted(tot.items()): print(k, "per launch %.4g"%(v/len(n[k])))
PY
