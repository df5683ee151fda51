#!/bin/bash
# round 3, visit D: whole GPU suite (with prints of the tightened tests), smoke, default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed\|^FAILED\|^E  .*assert\|^render_image_fast\|^update " $O/pytest_gpu.log | cut -c1-400 | tail -30
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default.json"))
r=d["roofline"]
print("fps", d["value"], "ms", d["ms_per_step"], "1-in-flight", d["one_frame_in_flight"], "samples/ray", d["samples_per_ray"], "cov", d["alpha_coverage"], "incomplete", d["frames_rerendered_eagerly"])
print("proc", d["procedural_track"])
print("roof", {k:r[k] for k in ("achieved","frac","traffic","traffic_source","avg_launch_us","solves","fetches_algorithm","fetches_loaded","kernel_resources","counters_source","ms_per_frame")})
print("hg", {k:v for k,v in d["hashgrid_lookup"].items() if k in ("frac","avg_launch_us","counters_source","frame_coherent")})
t=d["train"]; print("train", t.get("it_per_sec"), t.get("uniform_rays",{}).get("it_per_sec"), (t.get("refine") or {}).get("it_per_sec"), (t.get("refine") or {}).get("eager"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["frame_seconds"])
PY
