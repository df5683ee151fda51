#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, kernel-trace only) over a short eager bench run
# -> profiles/r01_pmc_traffic.json.  Each pass runs under its own timeout.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o r -- python $R/bench.py --steps 3 --warmup 2 --cpu-frames 0 --train-steps 0 --no-graph --no-profile > $O/pmc_$c.log 2>&1
done
python $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE/r_counter_collection.csv $O/pmc_WRITE_SIZE/r_counter_collection.csv $O/r01_pmc_traffic.json | tail -20
