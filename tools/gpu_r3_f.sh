#!/bin/bash
# in-flight sweep of the render bench
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for n in 1 2 3 4; do
  timeout 300 python bench.py --train-steps 0 --cpu-frames 0 --no-profile --in-flight $n --steps 200 > $O/bench_if$n.json 2> $O/bench_if$n.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_if$n.json')); print('in-flight $n', round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms', d['one_frame_in_flight'], 'incomplete', d['frames_rerendered_eagerly'])"
done
