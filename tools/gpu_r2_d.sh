#!/bin/bash
# Round-2 GPU visit D: full GPU suite (goldens), driver-protocol bench, kernel stats of the same command, PMC passes,
# L2 gather microbenchmark, encode A/Bs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
IA_WRITE_GOLDEN=1 timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 500 $O/bench_driver.json; echo; tail -2 $O/bench_driver.err
tools/ubench/l2gather > $O/l2gather.txt 2>&1; cat $O/l2gather.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_d && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d -o r -- python $R/bench.py --steps 20 --warmup 5 --cpu-frames 0 > $O/prof_d.log 2>&1 )
head -14 $O/prof_d/r_kernel_stats.csv | cut -c1-150
bash $R/tools/pmc_r2.sh > $O/pmc_r2.log 2>&1; grep -E "rc=" $O/pmc_r2.log
cd $R
bash tools/ab_field.sh "-DIA_ENC_S=8" "-DIA_ENC_S=2" "-DIA_ENC_MAX_WG_PER_XCD=512" "-DIA_ENC_S=4" 2>&1 | grep -E "===|uniform"
