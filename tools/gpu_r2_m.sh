#!/bin/bash
# bench (driver protocol, incl. the Morton-binned hash-grid figure) + kernel stats of a train-only run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 300 $O/bench_driver.json; echo
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_driver.json').read().strip().splitlines()[-1])
h=d['hashgrid_lookup']; print('random', h['Gsamples_per_s'], h['frac']); print('coherent', h.get('frame_coherent')); print('morton', h.get('morton_binned'))
print('train', d.get('train'))
PY
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_train && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o r -- python $R/bench.py --gpus 1 --train-only --steps 200 --warmup 10 > $O/prof_train.log 2>&1 )
head -40 $O/prof_train/r_kernel_stats.csv | cut -c1-100,200-
