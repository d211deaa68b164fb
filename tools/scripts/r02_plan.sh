#!/bin/bash
# pass plans of the 20-step run (2 pipelines, up to 10 batches per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for plan in "" "10,4,6" "7,7,6" "10,5,5" "6,10,4" "8,8,4"; do
  CASSNAT_PASS_PLAN=$plan timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plan [$plan]', d['value'], d['ms_per_step'])" || exit 1
done
done | tee gpurun_out/r03d_plans.txt
