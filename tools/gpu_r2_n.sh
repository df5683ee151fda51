#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_data.py tests/test_gpu_fit.py tests/test_gpu_training.py -q -m gpu -x 2>&1 | tail -12 | cut -c1-300
for i in 1 2; do timeout 200 python bench.py --train-only --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d['train']; print(round(t['it_per_sec'],1), 'it/s', t.get('launch_mode'), t['samples_candidates_last_step'], t['mse_last'])"; done
