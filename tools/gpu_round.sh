#!/bin/bash
# One GPU visit: the bit-exactness tests that guard the search / occupancy path, then a short bench (no CPU leg, no training).
#   usage (on the GPU box, through gpurun): bash tools/gpu_round.sh <tag> [extra pytest -k expression]
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out; mkdir -p $O; cd $R
tag=${1:-x}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_pin.py tests/test_gpu_fullconfig.py tests/test_gpu_edge_cases.py -x -q -m gpu > $O/t_${tag}.log 2>&1
tail -5 $O/t_${tag}.log | cut -c1-300
timeout 300 python bench.py --cpu-frames 0 --train-steps 0 > $O/bench_${tag}.json 2> $O/bench_${tag}.err
python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${tag}.json").read().strip().splitlines()[-1])
    print("fps", d["value"], "ms", d["ms_per_step"], "steady", d.get("value_steady"))
    r = d.get("roofline", {})
    print({k: r.get(k) for k in ("kernel", "achieved", "frac", "avg_us", "launches_per_step")})
    for k in ("search_launches", "kernels"):
        if k in d: print(k, json.dumps(d[k])[:600])
except Exception as e:
    print("bench parse failed", e); print(open("$O/bench_${tag}.err").read()[-1500:])
PY
