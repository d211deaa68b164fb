#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 20 --warmup 5 --exit-after-timed"
run() { echo -n "$* : "; timeout -k 10 200 python bench.py $B "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1; }
for rep in 1 2 3; do
run
run --streams 1 --coalesce 20
run --coalesce 14 --plan 12,8
run --coalesce 14 --plan 14,6
run --plan 10,5,5
run --plan 7,7,6
run --plan 5,5,5,5
run --streams 3 --plan 7,7,6
run --coalesce 16 --plan 16,4
done | tee gpurun_out/r04t_plan_sweep_20steps.txt
