#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/ab_build.sh ia_snarf.hip "-DIA_FETCH_GROUP=1" "-DIA_FETCH_GROUP=1 -DIA_SEARCH_NP=16 -DIA_SEARCH_THREADS=64" "-DIA_FETCH_GROUP=2" 2>&1 | grep -v warning | grep fps
