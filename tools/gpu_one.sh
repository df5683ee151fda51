#!/bin/bash
# one pytest selection on the GPU box: tools/gpu_one.sh <pytest args>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest "$@" -q -s > $O/one.log 2>&1; tail -60 $O/one.log
