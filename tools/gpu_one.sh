#!/bin/bash
# run a pytest selection on the GPU box:  tools/gpu_one.sh "<pytest args>"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest $1 -q -m gpu > $O/pytest_one.log 2>&1; tail -25 $O/pytest_one.log
