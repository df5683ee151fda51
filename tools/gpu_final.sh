#!/bin/bash
# Dress rehearsal of the driver's round-end sequence on one box + the evidence under profiles/:
#   build() -> default bench.py (the line the driver records) -> the same command under rocprofv3 --kernel-trace --stats
#   -> training kernel stats.   usage (through gpurun): bash tools/gpu_final.sh r05
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out; mkdir -p $O; cd $R
RND=${1:-r05}
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/${RND}_bench_driver.json 2> $O/${RND}_bench_driver.err
tail -3 $O/${RND}_bench_driver.err
( time python bench.py ) > $O/${RND}_bench_default.json 2> $O/${RND}_bench_default.err
tail -3 $O/${RND}_bench_default.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o r -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $O/${RND}_bench_prof.json 2> $O/${RND}_bench_prof.err
cp /tmp/pp/r_kernel_stats.csv $O/${RND}_bench_kernel_stats.csv 2>/dev/null; head -6 $O/${RND}_bench_kernel_stats.csv | cut -c1-120
cd $R; bash tools/prof_train_stats.sh 2>&1 | tail -12
python - <<PY
import json
for f in ("${RND}_bench_driver", "${RND}_bench_default"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(f, "fps", round(d["value"], 1), "ms", round(d["ms_per_step"], 3), "roof", r.get("kernel"), round(r.get("frac", 0), 3), "traffic", r.get("traffic"), (r.get("hbm") or {}).get("frac"))
        print("  cpu", json.dumps(d.get("cpu_baseline"))[:400])
        t = d.get("train") or {}
        print("  train", t.get("it_per_sec"), (t.get("uniform_rays") or {}).get("it_per_sec"), (t.get("refine") or {}).get("it_per_sec"))
        print("  hashgrid", {k: d["hashgrid_lookup"].get(k) for k in ("frac", "l2_request_rate_frac")}, (d["hashgrid_lookup"].get("frame_coherent") or {}).get("frac"))
    except Exception as e:
        print(f, "parse failed", e)
PY
