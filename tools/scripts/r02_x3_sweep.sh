#!/bin/bash
# pipelines x batches-per-pass sweep of the split-bf16 engine
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 1" "1 3" "1 6" "1 9" "2 3" "2 6" "3 3" "3 6" "2 9"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --streams $1 --coalesce $2 --steps 72 --warmup 9 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelines', $1, 'coalesce', $2, d['value'], d['ms_per_step'])" || exit 1
done 2>&1 | tee gpurun_out/r02q_x3_sweep.txt
