#!/bin/bash
# shared by the tools/ab_*.sh helpers: rebuild ONE translation unit with extra flags through instantavatar_amd/build.py
# (the variant's flags enter that unit's hashes in the library manifest, so a variant can never pass for the default build;
# `ab_rebuild <file> ""` restores the default).   usage:  source tools/ab_lib.sh; ab_rebuild ia_snarf.hip "-DIA_X=1"
# (A/B runs execute variants on purpose: _lib.lib() would refuse a library whose manifest differs from the default checkout)
export IA_ALLOW_STALE_LIB=1
ab_rebuild() {
  local f=$1 flags=$2
  if [ -n "$flags" ]; then IA_EXTRA_HIPCC_FLAGS="$f:$flags" python instantavatar_amd/build.py > /dev/null || return 1
  else python instantavatar_amd/build.py > /dev/null || return 1; fi
}
ab_flags_env() {  # the environment a python process needs so that build.needs_build() agrees with the variant it runs on
  if [ -n "$2" ]; then echo "IA_EXTRA_HIPCC_FLAGS=$1:$2"; else echo "IA_EXTRA_HIPCC_FLAGS="; fi
}
