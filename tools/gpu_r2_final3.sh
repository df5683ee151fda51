#!/bin/bash
# closing measurements (second half of round 2): GPU suite, smoke, driver-protocol bench, kernel stats of the same command and of a train-only run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 200 $O/bench_driver.json; echo
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_final && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o r -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > $O/prof_final.log 2>&1 )
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_train && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o r -- python $R/bench.py --gpus 1 --train-only --steps 200 --warmup 10 > $O/prof_train.log 2>&1 )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/atomics.hip -o /tmp/atomics && /tmp/atomics > $O/ubench_atomics.txt
ls $O/prof_final $O/prof_train | head
