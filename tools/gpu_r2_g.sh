#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/kstat.sh 2>&1 | grep -E "k_occ|k_precompute|k_march"
for i in 1 2; do timeout 200 python bench.py --steps 40 --warmup 5 --cpu-frames 0 --train-steps 0 --no-profile 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'fps',round(d['ms_per_step'],3),'ms')"; done
