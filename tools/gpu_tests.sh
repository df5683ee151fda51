#!/bin/bash
# GPU suite only (writes the reference goldens into gpurun_out/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
IA_WRITE_GOLDEN=1 timeout 1200 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
grep -E "^(512|1024)|'step'|voxelised|product init" $O/pytest_gpu.log | head -20
