#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for mode in coherent random; do
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  d=pmcq_${mode}_$(echo $c | tr ' ' '_')
  rm -rf $O/$d
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$d -o r -- python $R/tools/pmc_encode.py $mode > $O/$d.log 2>&1
done
done
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out"
for mode in ("coherent","random"):
    n=None
    for f in glob.glob(O+"/pmcq_%s_*.log"%mode):
        for line in open(f):
            if line.startswith("frame-coherent samples:"): n=int(line.split(":")[1])
    if mode=="random": n=1<<20
    for f in sorted(glob.glob(O+"/pmcq_%s_*/**/*counter_collection.csv"%mode, recursive=True)):
        rows=[r for r in csv.DictReader(open(f)) if "k_encode_xcd" in r["Kernel_Name"]]
        ids=sorted({int(r["Dispatch_Id"]) for r in rows})[-4:]
        acc=collections.defaultdict(float); us=[]
        for r in rows:
            if int(r["Dispatch_Id"]) in ids:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"])/len(ids)
                us.append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
        us.sort()
        print(mode, "samples", n, "us", us[len(us)//2] if us else None, {k: "%.4g (%.2f/sample)"%(v, v/n) for k,v in acc.items()})
PY
