#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg"
for rep in 1 2; do
for st in 0 2 4 8 12; do
  CASSNAT_CHAIN_STAGGER=$st timeout -k 10 200 python bench.py $B --steps 200 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('stagger $st:', d['value'], d['ms_per_step'], 'chain frac', r['frac'], 'us', r['avg_launch_us'], 'alone', r['isolated_at_width_frac'])"
done
done | tee gpurun_out/r03l_stagger.txt
