#!/bin/bash
# A/B helper (dev tool): rebuild with extra -D flags for ia_snarf.hip and time the compact search alone
for flags in "$@"; do
  ( cd instantavatar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -x hip -c ia_snarf.hip -o ia_snarf.hip.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinstantavatar_hip.so *.o )
  timeout 120 python tools/bench_search.py "$flags" 16 2>&1 | tail -1
  timeout 120 python tools/bench_search.py "$flags" 4 2>&1 | tail -1
done
