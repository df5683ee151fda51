#!/bin/bash
# One GPU-box visit: tests, smoke, default bench, rocprofv3 kernel stats of the same bench command.
# Every step runs under its own timeout (a hung profiler pass must not eat the GPU budget).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
timeout 400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 600 $O/bench_default.json; echo
if [ "$1" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof_final
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r -- python $R/bench.py --cpu-frames 0 > $O/prof_final.log 2>&1
  ls $O/prof_final
fi
