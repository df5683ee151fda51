#!/bin/bash
# round 3, visit C: quad-cooperative k_search -- parity first, then A/B timings of the compact search alone
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ref_pin.py tests/test_gpu_edge_cases.py -q -x > $O/one.log 2>&1; tail -4 $O/one.log
rm -f /tmp/bench_search_pts_*.pt
bash tools/ab_search.sh "-DIA_SEARCH_QUAD=0" "-DIA_SEARCH_QUAD=1" "-DIA_QUAD_GROUP=4" "-DIA_QUAD_GROUP=2" "-DIA_SEARCH_WAVES_PER_EU=4" "-DIA_QUAD_GROUP=4 -DIA_SEARCH_WAVES_PER_EU=4" 2>&1 | grep -v warning | tee $O/ab_search.txt
