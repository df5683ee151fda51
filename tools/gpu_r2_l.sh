#!/bin/bash
cd $GRAFT_REPO_ROOT
for fl in "-DIA_MARCH_LDS=0" "-DIA_MARCH_LDS=1"; do
  ( cd instantavatar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $fl -x hip -c ia_render.hip -o ia_render.hip.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libinstantavatar_hip.so *.o )
  for i in 1 2; do timeout 200 python bench.py --steps 40 --warmup 5 --cpu-frames 0 --train-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[$fl]', round(d['value'],1), 'fps', d.get('one_frame_in_flight'))"; done
done
