#!/bin/bash
# closing visit: GPU suite, smoke, the default bench (driver protocol AND default), rocprofv3 kernel stats of the same command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 300 $O/bench_driver.json; echo
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; head -c 300 $O/bench_default.json; echo
timeout 300 python bench.py --gpus 1 --train-only --force-collectives --steps 100 --warmup 5 > $O/bench_train_collectives.json 2> $O/bench_train_collectives.err; head -c 1200 $O/bench_train_collectives.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_final
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r -- python $R/bench.py --cpu-frames 0 > $O/prof_final.log 2>&1
cp $O/prof_final/r_kernel_stats.csv $O/r03_bench_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_final
head -5 $O/r03_bench_kernel_stats.csv | cut -c1-160
