#!/bin/bash
# round 3: quad-cooperative k_search -- A/B timings of the compact search alone, then parity with the variant given as $1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f /tmp/bench_search_pts_*.pt
bash tools/ab_search.sh "-DIA_SEARCH_QUAD=0" "$@" 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^" | tee $O/ab_search.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ref_pin.py tests/test_gpu_edge_cases.py -q -x > $O/one.log 2>&1; tail -3 $O/one.log
