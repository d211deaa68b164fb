#!/bin/bash
# round 3: how the 20 timed steps are cut into engine passes (workgroup-count quantisation: a chain launch has rows / 128
# workgroups on 256 CUs - 8 batches = 500 workgroups = two full rounds, 10 batches = 625 = 2.44)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps $1 --warmup 5 --streams $2 --coalesce $3 ${4:+--plan $4} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps $1 streams $2 coalesce $3 plan ${4:-equal}:', d['value'], d['ms_per_step'], 'chain frac', r['frac'], 'us', r['avg_launch_us'], 'alone', r['isolated_at_width_frac'])"; }
for rep in 1 2; do
run 20 2 10
run 20 2 10 8,8,4
run 20 2 8 8,4,8
run 20 2 4
run 20 1 20
run 20 1 16 16,4
run 20 1 8 8,8,4
run 20 3 8 8,8,4
done 2>&1 | tee gpurun_out/r03d_plans.txt
run 200 2 10
run 200 2 8
run 200 1 8
run 200 1 16
run 200 3 8
