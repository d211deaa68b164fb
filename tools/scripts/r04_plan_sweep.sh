#!/bin/bash
# the driver's 20-step run: how the 20 batches are cut into engine passes (and how many pipelines take them)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "2 10,10" "2 12,8" "2 8,12" "2 6,6,4,4" "2 5,5,5,5" "2 7,7,3,3" "2 8,8,2,2" "3 7,7,6" "3 8,6,6" "3 5,5,5,5" "2 4,8,8" "2 9,9,1,1"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 20 --warmup 5 --streams $1 --coalesce 12 --plan $2 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $1 plan $2', d['value'], d['ms_per_step'])" || echo "streams $1 plan $2 failed"
done
done | tee gpurun_out/r04e_plan_sweep_20steps.txt
