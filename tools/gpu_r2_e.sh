#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_density_update.py -q -m gpu > $O/pytest_e.log 2>&1; tail -4 $O/pytest_e.log
timeout 300 python tools/bench_hgbwd.py 2>&1 | grep -E "all 16|level  [0-7]"
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-profile --train-steps 200 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fps', round(d['value'],1), 'train', d['train'])"
