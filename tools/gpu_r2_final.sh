#!/bin/bash
# Round-2 closing visit: full GPU suite (also under the other tcnn level-3 layout), smoke, driver-protocol bench,
# default bench, rocprofv3 kernel stats of the driver command.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
IA_TCNN_LEVEL3_RES=55 timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu_res55.log 2>&1; tail -3 $O/pytest_gpu_res55.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 400 $O/bench_driver.json; echo; tail -2 $O/bench_driver.err
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 300 $O/bench_default.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_final && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $O/prof_final.log 2>&1 )
head -12 $O/prof_final/r_kernel_stats.csv | cut -c1-140
