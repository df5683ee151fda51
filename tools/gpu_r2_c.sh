#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_density_update.py tests/test_gpu_fullconfig.py -q -m gpu -k "occupancy or density or voxelise or 512" > $O/pytest_c.log 2>&1; tail -4 $O/pytest_c.log
bash tools/kstat.sh 2>&1 | grep -E "k_occ|k_precompute|k_march|k_search|k_encode|k_composite|k_probe"
bash tools/ab_kstat.sh ia_snarf.hip "k_precompute" "-DIA_PRE_UNROLL=4" "-DIA_PRE_UNROLL=6" "-DIA_PRE_UNROLL=8" "-DIA_PRE_UNROLL=12" "-DIA_PRE_VPT=2 -DIA_PRE_UNROLL=8" "-DIA_PRE_UNROLL=2"
