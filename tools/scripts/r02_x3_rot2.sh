#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rot in 1 0; do
  export CASSNAT_FFN_X3_ROTATE=$rot
  for cfg in "1 1" "1 6" "3 3"; do
    set -- $cfg
    timeout -k 10 200 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --streams $1 --coalesce $2 --steps 72 --warmup 9 2>/dev/null | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rotate', $rot, 'pipelines', $1, 'coalesce', $2, d['value'], d['ms_per_step'], 'ffn ms/batch', d['stage_ms'].get('ffn_fused_x3'))" || exit 1
  done
done 2>&1 | tee gpurun_out/r02q_x3_rot.txt
