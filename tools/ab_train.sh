#!/bin/bash
# A/B helper (dev tool): rebuild one source with extra -D flags and run the train-only bench
f=$1; shift
for flags in "$@"; do
  ( cd instantavatar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $flags -x hip -c $f -o $f.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinstantavatar_hip.so *.o )
  for i in 1 2; do timeout 200 python bench.py --train-only --steps 300 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d['train']; print('[$flags]', round(t['it_per_sec'],1), 'it/s', t.get('launch_mode'), t['mse_last'])"; done
done
