#!/bin/bash
# Round-2 GPU visit A: full GPU suite (writes the reference goldens), smoke, bench under the driver protocol,
# kernel trace + stats of a short bench run.  Every step under its own timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
IA_WRITE_GOLDEN=1 timeout 900 python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
grep -E "^(512|1024)|step'|voxelised" $O/pytest_gpu.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 1500 $O/bench_driver.json; echo; tail -3 $O/bench_driver.err
if [ "$1" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof_a
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -o r -- python $R/bench.py --steps 20 --warmup 5 --cpu-frames 0 --train-steps 0 --spinup-max-ms 300 > $O/prof_a.log 2>&1
  ls -la $O/prof_a | head; tail -2 $O/prof_a.log | head -c 600
fi
