#!/bin/bash
# closing measurements: GPU suite, driver-protocol bench, kernel stats of the same command, PMC passes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 300 $O/bench_driver.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_final && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $O/prof_final.log 2>&1 )
head -8 $O/prof_final/r_kernel_stats.csv | cut -c1-120
bash $R/tools/pmc_r2.sh > $O/pmc_r2.log 2>&1; grep -E "rc=" $O/pmc_r2.log
