#!/bin/bash
# round 6: cache policy of the hashed-level gathers of k_encode_xcd -- default (L1-allocating) / nt / sc1 per
# level group (hashed 4..7 = LO, 8..15 = HI) -- time on random + frame-coherent samples, feature checksum, and the vector-L1
# counters of the frame-coherent launch.   usage (on the box): bash tools/ab_encode_policy.sh [pmc]
source "$(dirname "$0")/ab_lib.sh"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/enc_policy
for flags in "" "-DIA_ENC_POL_H_HI=1" "-DIA_ENC_POL_H_HI=2" "-DIA_ENC_POL_H_HI=1 -DIA_ENC_POL_H_LO=1" "-DIA_ENC_POL_H_HI=2 -DIA_ENC_POL_H_LO=2" "-DIA_ENC_POL_H_LO=1"; do
  cd $R; ab_rebuild ia_field.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  echo "=== ia_field.hip [$flags]"
  env $(ab_flags_env ia_field.hip "$flags") timeout 200 python $R/tools/ab_encode_policy.py 2>&1 | grep -E "random|coherent"
  if [ "$1" = "pmc" ]; then
    tag=$(echo "$flags" | tr -c 'A-Za-z0-9=\n' '_'); i=0
    for c in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
      i=$((i+1)); d=/tmp/pmcq_$i; rm -rf $d
      (cd /tmp && export TMPDIR=/tmp && env $(ab_flags_env ia_field.hip "$flags") timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o r -- python $R/tools/pmc_encode.py coherent > $d.log 2>&1)
      python $R/tools/pmc_condense.py $(dirname $(find $d -name "r_counter_collection.csv" | head -1)) $O/enc_policy/v${tag}_$i.csv 2>/dev/null
      grep "k_encode_xcd" $O/enc_policy/v${tag}_$i.csv
    done
  fi
done
cd $R; ab_rebuild ia_field.hip ""
