#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/ab_build.sh ia_snarf.hip "-DIA_FETCH_SKIP_OUTSIDE=0" "-DIA_FETCH_SKIP_OUTSIDE=1" 2>&1 | grep fps
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_ref_pin.py -q -m gpu -x 2>&1 | tail -2
