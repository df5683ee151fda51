#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 python tools/bench_precompute.py ws 2>&1 | tail -1
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_driver.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['one_frame_in_flight'], d['train'])
PY
