#!/bin/bash
# usage: tools/gpu_py.sh <shell command line>   (runs it from the repo root on the GPU box, output in gpurun_out/py_tool.log)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash -c "$*" > gpurun_out/py_tool.log 2>&1; tail -30 gpurun_out/py_tool.log | cut -c1-400
