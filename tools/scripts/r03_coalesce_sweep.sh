#!/bin/bash
# steady-state headline against batches per engine pass and decode pipelines (240 steps: a whole number of passes for each width)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for s in 2 3; do
  for c in 6 8 10 12 16; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 240 --warmup 5 --coalesce $c --streams $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s coalesce $c', d['value'], d['ms_per_step'], d['roofline']['frac'])" || exit 1
  done
done
done | tee gpurun_out/r03r_coalesce_sweep.txt
