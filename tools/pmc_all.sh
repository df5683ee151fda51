#!/bin/bash
# Every PMC pass the bench line cites, on the library as it is NOW, summarised with the hashes of its device code inside
# (tools/pmc_condense_all.py -> gpurun_out/profiles_$RND/${RND}_pmc_*.json; copy those to profiles/).
# One counter set per pass, --kernel-trace only (MI355X_MICROARCH.md; gpurun refuses --pmc combined with other traces).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; RND=${IA_PMC_ROUND:-r05}; export IA_PMC_ROUND=$RND
run() {  # run <dir> <counter set> -- <command...>
  local d=$1 c=$2; shift 3
  rm -rf $O/$d
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$d -o r -- "$@" > $O/$d.log 2>&1
  echo "$d rc=$? $(ls $O/$d 2>/dev/null | tr '\n' ' ')"
}
BENCH="python $R/bench.py --steps 3 --warmup 2 --cpu-frames 0 --train-steps 20 --no-graph --no-profile --spinup-max-ms 50"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  run pmc2_$(echo $c | tr ' ' '+') "$c" -- $BENCH
done
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  run pmc_enc_$(echo $c | tr ' ' '_') "$c" -- python $R/tools/pmc_encode.py
done
for c in "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  run pmc_encc_$(echo $c | tr ' ' '_') "$c" -- python $R/tools/pmc_encode.py coherent
done
for c in "TCC_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum"; do
  run pmc3_$(echo $c | tr ' ' '+') "$c" -- python $R/bench.py --train-only --steps 30 --warmup 5 --no-graph
done
python $R/tools/pmc_condense_all.py $O $O/profiles_$RND > $O/pmc_all_summary.txt 2>&1; tail -5 $O/pmc_all_summary.txt
# the raw per-dispatch CSVs are tens of MB per pass (gpurun copies back at most 64 MiB): keep one condensed row per
# (kernel, counter) of every pass next to the summaries and drop the raw files
mkdir -p $O/profiles_$RND/${RND}_pmc
for d in $O/pmc2_* $O/pmc_enc_* $O/pmc_encc_* $O/pmc3_*; do
  [ -d "$d" ] || continue
  python $R/tools/pmc_condense.py $d $O/profiles_$RND/${RND}_pmc/$(basename $d).csv 2>/dev/null
  rm -rf $d
done
rm -rf $O/pmcq_* $O/pmc_s[0-9]* $O/prof_frame $O/prof_refine $O/pmc_records 2>/dev/null
ls -la $O/profiles_$RND $O/profiles_$RND/${RND}_pmc | head -40; du -sh $O
