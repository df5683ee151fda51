#!/bin/bash
# run a python dev tool on the GPU box:  tools/gpu_py.sh tools/<script>.py [args]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python "$@" 2>&1 | tee gpurun_out/py_tool.log | tail -60
