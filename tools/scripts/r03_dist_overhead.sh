#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for steps in 20 200; do
for mode in "" "--force-dist"; do
  timeout -k 10 300 python bench.py $mode --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps $steps --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $steps mode [$mode]', d['value'], d['ms_per_step'], d.get('weight_broadcast_ms'))" || exit 1
done
done | tee gpurun_out/r03y_dist_overhead.txt
