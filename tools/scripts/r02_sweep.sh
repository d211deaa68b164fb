#!/bin/bash
# pipelines x batches-per-pass sweep of the headline, at the driver's step count and at the default one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4 1" "4 2" "3 2" "2 2" "2 4" "3 3" "3 4" "4 4" "2 8" "1 8"; do
  set -- $cfg
  for steps in 20 200; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --streams $1 --coalesce $2 --steps $steps --warmup 5 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelines', $1, 'coalesce', $2, 'steps', $steps, d['value'], d['ms_per_step'], d['roofline']['frac'])" || exit 1
  done
done 2>&1 | tee gpurun_out/r02b_sweep.txt
