#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
( cd /tmp && export TMPDIR=/tmp && rm -rf $O/prof_train && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o r -- python $R/bench.py --gpus 1 --train-only --steps 200 --warmup 10 > $O/prof_train.log 2>&1 )
tail -1 $O/prof_train.log | cut -c1-200
