#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_ref_pin.py -q -m gpu -k "precompute or frozen or production" 2>&1 | tail -2
bash tools/ab_kstat.sh ia_snarf.hip "k_precompute" "-DIA_PRE_GROUP=8" "-DIA_PRE_GROUP=4" "-DIA_PRE_GROUP=6" "-DIA_PRE_GROUP=12" "-DIA_PRE_GROUP=8 -DIA_PRE_VPT=2" "-DIA_PRE_GROUP=12 -DIA_PRE_VPT=2" "-DIA_PRE_GROUP=24 -DIA_PRE_VPT=2" "-DIA_PRE_GROUP=24 -DIA_PRE_VPT=1"
