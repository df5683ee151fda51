#!/bin/bash
# round 3, visit A: the new refine tests first, then the whole GPU suite, smoke, the default bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_refine.py -q -s > $O/refine.log 2>&1; tail -25 $O/refine.log
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 400 $O/bench_default.json; echo
