#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "pipelined or graph" 2>&1 | tail -15
for n in 2 1 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --train-steps 0 --in-flight $n 2> $O/bk.err | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('in_flight', d['frames_in_flight'], round(d['value'],1), 'fps', round(d['ms_per_step'],3), 'ms  first', round(d['value_first_window'],1), 'steady', round(d['value_steady'],1), d['one_frame_in_flight'], d['launch_mode'], 'rerendered', d['frames_rerendered_eagerly'], 'spr', round(d['samples_per_ray'],4))"; tail -2 $O/bk.err; done
