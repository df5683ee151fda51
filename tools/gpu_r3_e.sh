#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fullconfig.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -q -x > $O/one.log 2>&1; tail -3 $O/one.log
timeout 600 python bench.py --train-steps 0 --cpu-frames 0 > $O/bench_render.json 2> $O/bench_render.err; tail -2 $O/bench_render.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_render.json"))
print("fps", d["value"], "ms", d["ms_per_step"], "1-in-flight", d["one_frame_in_flight"], "samples/ray", d["samples_per_ray"], "cov", d["alpha_coverage"], "incomplete", d["frames_rerendered_eagerly"])
print("proc", d["procedural_track"])
PY
