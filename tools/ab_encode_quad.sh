#!/bin/bash
# round 6: k_encode_xcd with the aligned 16-byte GROUP of four entries per (y, z) pair (IA_ENC_QUAD=1) instead of the aligned pair:
# one dwordx4 gather serves both x-corners for 3 lanes in 4 -- time on random + frame-coherent samples, feature checksum (must be
# identical), per samples-per-thread setting.   usage (on the box): bash tools/ab_encode_quad.sh
source "$(dirname "$0")/ab_lib.sh"
R=$GRAFT_REPO_ROOT
for flags in "-DIA_ENC_QUAD=0" "-DIA_ENC_QUAD=1 -DIA_ENC_S=4" "" "-DIA_ENC_QUAD=1 -DIA_ENC_S=2" "-DIA_ENC_QUAD=0 -DIA_ENC_S=3" "-DIA_ENC_QUAD=0 -DIA_ENC_S=2"; do
  cd $R; ab_rebuild ia_field.hip "$flags" || { echo "build failed: [$flags]"; continue; }
  echo "=== ia_field.hip [$flags]"
  env $(ab_flags_env ia_field.hip "$flags") timeout 200 python $R/tools/ab_encode_policy.py 2>&1 | grep -E "random|coherent"
done
cd $R; ab_rebuild ia_field.hip ""
