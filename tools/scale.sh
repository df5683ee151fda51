#!/bin/bash
# Scaling runs on one node (dev tool; the driver's SCALE_rNN.json is the authoritative curve): N = 1, 2, 4, 8 ranks, EVERY N
# -- N = 1 included -- through the same launcher the driver uses (torch.distributed.run on 127.0.0.1), first the inference
# line (frames sharded round-robin, no data-path collective), then `--train-only` (RCCL gradient all-reduce; the eager
# comparison reports the exposed all-reduce time per step).  Every rank echoes what it saw on stderr ([bench rank r/N] ...),
# rank 0's JSON carries the same records under "ranks".
#   usage: tools/scale.sh [max_gpus] [steps] [out_dir]        e.g. tools/scale.sh 8 50 gpurun_out/scale
MAXN=${1:-8}; STEPS=${2:-50}; OUT=${3:-gpurun_out/scale}
R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
port=29500
for n in 1 2 4 8; do
  [ $n -le $MAXN ] || break
  if [ $n -gt $HAVE ]; then echo "N=$n: only $HAVE device(s) here, skipped"; continue; fi
  for mode in render train tile; do
    port=$((port + 1))
    extra=""; [ $mode = train ] && extra="--train-only"; [ $mode = tile ] && extra="--tile-shard"
    log=$OUT/${mode}_n$n
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        $R/bench.py --gpus $n --steps $STEPS --warmup 5 --cpu-frames 0 --train-steps 0 $extra > $log.json 2> $log.err
    python - "$log.json" $n $mode <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("ranks") or (d.get("train") or {}).get("ranks") or []
    ex = ((d.get("train") or {}).get("eager") or {}).get("exposed_allreduce_ms_per_step")
    print("N=%s %-6s %s = %.1f %s  (%.3f ms/step)  ranks seen: %s  exposed all-reduce: %s" % (
        sys.argv[2], sys.argv[3], d["metric"], d["value"], d["unit"], d["ms_per_step"], sorted({x["world_size_seen"] for x in r}), ex))
except Exception as e:
    print("N=%s %s: no result line (%s); see the .err file" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
