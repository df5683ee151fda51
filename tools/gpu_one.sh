#!/bin/bash
# usage: tools/gpu_one.sh <pytest args>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest "$@" > gpurun_out/pytest_one.log 2>&1; tail -40 gpurun_out/pytest_one.log | cut -c1-300
