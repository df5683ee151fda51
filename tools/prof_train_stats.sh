#!/bin/bash
# rocprofv3 --kernel-trace --stats of the two training workloads of the bench line (config 2: patch sampler; config 4: refinement)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for mode in plain refine; do
  rm -rf $O/prof_$mode
  arg=""; [ $mode = plain ] && arg="--plain"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -o r -- python $R/tools/prof_refine.py 100 $arg > $O/prof_$mode.log 2>&1
  grep it_per_sec $O/prof_$mode.log
  cp $O/prof_$mode/r_kernel_stats.csv $O/${IA_PMC_ROUND:-r06}_train_${mode}_kernel_stats.csv; rm -rf $O/prof_$mode
  head -8 $O/${IA_PMC_ROUND:-r06}_train_${mode}_kernel_stats.csv | cut -c1-110
done
