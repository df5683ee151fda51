#!/bin/bash
# Issue-level counters of k_search on a frame's sample points (tools/bench_search.py), one counter set per pass:
# which instruction class occupies the SIMDs, and how many lanes a VALU instruction carries.  Summaries (one row per kernel
# and counter) -> gpurun_out/pmc_issue/*.csv ; the list of available SQ counters -> gpurun_out/pmc_issue/avail.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_issue; mkdir -p $O
timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u > $O/avail.txt; wc -l $O/avail.txt
i=0
for c in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
         "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/p$i -o r -- env PYTHONPATH=$R python $R/tools/bench_search.py pmc 16 > $O/p$i.log 2>&1
  echo "pass $i rc=$? [$c]"
  f=$(find $O/p$i -name "r_counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_condense.py $(dirname $f) $O/p$i.csv && grep "k_search" $O/p$i.csv; fi
  rm -rf $O/p$i
done
