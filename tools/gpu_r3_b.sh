#!/bin/bash
# round 3, visit B: refine tests (incl. graph replay), training tests, data tests, fit test, then the bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_refine.py tests/test_gpu_training.py tests/test_gpu_fit.py tests/test_gpu_data.py -q -s > $O/one.log 2>&1; grep -n "^step\|^eager\|passed\|failed\|Error\|^E  .*assert" $O/one.log | cut -c1-300 | tail -30
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default.json"))
print(d["value"], d["ms_per_step"])
t=d.get("train",{})
print({k:t.get(k) for k in ("it_per_sec","launch_mode","graph_capture_error")})
print(json.dumps(t.get("refine"), indent=1))
PY
