#!/bin/bash
cd $GRAFT_REPO_ROOT
for fl in "-DIA_HGB_LEVEL_UNROLL=1" "-DIA_HGB_LEVEL_UNROLL=2" "-DIA_HGB_LEVEL_UNROLL=4"; do
  ( cd instantavatar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $fl -x hip -c ia_field.hip -o ia_field.hip.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinstantavatar_hip.so *.o )
  timeout 200 python tools/bench_hgbwd_patch.py "$fl" 2>&1 | grep -E "all 16"
  timeout 200 python bench.py --train-only --steps 200 --warmup 10 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); t=d['train']; print('[$fl]', round(t['it_per_sec'],1), 'it/s')"
done
