#!/bin/bash
# dev tool: kernel trace of a training-only bench run -> per-step breakdown
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; rm -rf $O/prof_train
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/prof_train -o r -- python $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --train-steps 40 > $O/prof_train.log 2>&1
python - <<'PY'
import csv, collections, os, glob
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_train/**/*kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'k_march_train_compact' in n]
a,b=idx[10],idx[30]
t0=int(rows[a]['Start_Timestamp']); t1=int(rows[b]['Start_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[a:b])
print("wall per step %.3f ms, gpu busy per step %.3f ms, launches per step %.1f"%((t1-t0)/20/1e6, busy/20/1e6, (b-a)/20))
d=collections.defaultdict(lambda:[0,0.0])
for r in rows[a:b]:
    n=r['Kernel_Name'].split('(')[0][:70]; d[n][0]+=1; d[n][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,v in sorted(d.items(),key=lambda x:-x[1][1])[:48]:
    print(f"{k:70s} {v[0]/20:6.1f}/step {v[1]/20:8.1f} us/step")
PY
