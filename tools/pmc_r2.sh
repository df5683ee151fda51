#!/bin/bash
# Round-2 PMC passes (one counter set per pass, --kernel-trace only, as the micro-architecture guide prescribes)
# over a short EAGER bench run that also trains 20 steps (so the backward kernels are seen).
#   bash tools/pmc_r2.sh            -> gpurun_out/pmc2_<set>/...  + summaries printed and written by tools/pmc_r2.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
CMD="python $R/bench.py --steps 3 --warmup 2 --cpu-frames 0 --train-steps 20 --no-graph --no-profile --spinup-max-ms 50"
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
      "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
      "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
      "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
      "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_REQ_sum TCC_EA0_RDREQ_sum")
for c in "${SETS[@]}"; do
  n=$(echo $c | tr ' ' '+')
  rm -rf $O/pmc2_$n
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc2_$n -o r -- $CMD > $O/pmc2_$n.log 2>&1
  echo "$n rc=$? $(ls $O/pmc2_$n 2>/dev/null | tr '\n' ' ')"
done
python $R/tools/pmc_r2.py $O $O
