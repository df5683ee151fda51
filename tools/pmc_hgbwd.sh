#!/bin/bash
# L2 atomic / request counters of the hash-grid backward in a short eager train-only run (patch workload)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --list-avail 2>/dev/null | grep -iE "TCC_ATOMIC|TCC_REQ|TCC_WRITE|TCC_EA0_ATOMIC|TCC_EA0_WRREQ" | head -20 > $O/avail_atomic.txt; cat $O/avail_atomic.txt | cut -c1-160
CMD="python $R/bench.py --train-only --steps 30 --warmup 5 --no-graph"
for c in "TCC_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $c | tr ' ' '+')
  rm -rf $O/pmc3_$n
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc3_$n -o r -- $CMD > $O/pmc3_$n.log 2>&1
  echo "$n rc=$? $(ls $O/pmc3_$n 2>/dev/null | tr '\n' ' ')"
done
python - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out'
for d in sorted(glob.glob(O+'/pmc3_*')):
    if not os.path.isdir(d): continue
    f=glob.glob(d+'/*counter_collection.csv')
    if not f: print(d,'no counter file'); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name'].split('(')[0][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k in agg:
        if 'hashgrid_bwd' in k or 'k_search' in k:
            print(os.path.basename(d), k, {c: round(v/cnt[(k,c)]) for c,v in agg[k].items()}, 'launches', max(cnt[(k,c)] for c in agg[k]))
PY
