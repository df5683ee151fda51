#!/bin/bash
# soak: a long render run and long training runs (plain / refine) -- no incomplete frames, no overflow, no non-finite skips, loss goes down
R=$GRAFT_REPO_ROOT; cd $R
timeout 300 python bench.py --steps 3000 --warmup 20 --train-steps 0 --cpu-frames 0 --no-profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('render 3000 frames:', round(d['value'],1), 'fps, incomplete', d['frames_rerendered_eagerly'], 'samples/ray', round(d['samples_per_ray'],3))"
timeout 300 python tools/prof_refine.py 1500 2>&1 | tail -1
timeout 300 python tools/prof_refine.py 1500 --plain 2>&1 | tail -1
