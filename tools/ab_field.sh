#!/bin/bash
# A/B helper (dev tool): rebuild ia_field.hip with extra -D flags on the GPU box, run tools/bench_field.py
for flags in "$@"; do
  ( cd instantavatar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -x hip -c ia_field.hip -o ia_field.hip.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinstantavatar_hip.so *.o )
  echo "=== [$flags]"
  python tools/bench_field.py 2>&1 | grep -E "uniform.*(65536|262144|1048576)"
done
