#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nw in 0 4 8; do
  for rep in 1 2; do
    CASSNAT_ATTN_NW=$nw timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_nw', $nw, d['value'], d['ms_per_step'])" || exit 1
  done
done 2>&1 | tee gpurun_out/r02g_attn_nw.txt
timeout -k 10 300 python tools/time_esa.py --reps 12 --same-seed 2>&1 | tail -2 | cut -c1-600
timeout -k 10 300 python tools/esa_phases.py 2>&1 | tail -9
