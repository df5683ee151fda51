#!/bin/bash
# dev tool: rebuild ia_snarf.hip with flags and print the probe / render split of k_search
for flags in "$@"; do
  ( cd instantavatar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -x hip -c ia_snarf.hip -o ia_snarf.hip.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinstantavatar_hip.so ia_error.cpp.o ia_snarf.hip.o ia_field.hip.o ia_render.hip.o ia_prof.hip.o ia_voxelise.hip.o ia_loss.hip.o ia_smpl_nn.hip.o )
  echo "=== [$flags]"; timeout 100 python tools/probe_stats.py 2>&1 | tail -2 | cut -c1-75
done
