#!/bin/bash
# round 5: the L1 miss-path counters of the isolated encoder on ONE frame's samples in three orders (pipeline / Morton / shuffled)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/enc_orders
for order in coherent coherent-morton coherent-shuffled; do
  i=0
  for c in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    i=$((i+1)); d=$O/pmcq_${order}_$i; rm -rf $d
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o r -- python $R/tools/pmc_encode.py $order > $d.log 2>&1
    python $R/tools/pmc_condense.py $d $O/enc_orders/${order}_$i.csv 2>/dev/null; rm -rf $d
    grep "k_encode_xcd" $O/enc_orders/${order}_$i.csv
  done
done
