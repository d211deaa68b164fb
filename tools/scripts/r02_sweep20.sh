#!/bin/bash
# pipelines x batches-per-pass at 200 steps (and 20)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "2 10" "2 8" "2 12" "2 16" "2 10"; do
  set -- $cfg
  for steps in 200; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --streams $1 --coalesce $2 --steps $steps --warmup 5 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelines', $1, 'coalesce', $2, 'steps', $steps, d['value'], d['ms_per_step'], d['roofline']['frac'])" || exit 1
  done
done 2>&1 | tee gpurun_out/r03i_sweep.txt
