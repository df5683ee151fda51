#!/bin/bash
# Round-2 GPU visit B: GPU suite, driver-protocol bench, kernel stats, then the PMC passes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
IA_WRITE_GOLDEN=1 timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 700 $O/bench_driver.json; echo; tail -2 $O/bench_driver.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_b
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o r -- python $R/bench.py --steps 20 --warmup 5 --cpu-frames 0 --spinup-max-ms 300 > $O/prof_b.log 2>&1
head -25 $O/prof_b/r_kernel_stats.csv | cut -c1-160
if [ "$1" = "pmc" ]; then bash $R/tools/pmc_r2.sh 2>&1 | tail -150; fi
