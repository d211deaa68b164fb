#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 20 --warmup 5 --streams $1 --coalesce $2 ${3:+--plan $3} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('streams $1 coalesce $2 plan ${3:-equal}:', d['value'], d['ms_per_step'], 'chain frac', r['frac'])"; }
for rep in 1 2 3; do
run 2 10
run 2 10 5,10,5
run 2 10 4,8,8
run 2 10 6,8,6
run 2 10 7,7,6
run 2 10 3,10,7
run 3 10 7,7,6
run 3 10 8,6,6
done 2>&1 | tee gpurun_out/r03m_plans20.txt
