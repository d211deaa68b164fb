#!/bin/bash
# fp8 engine: pipelines x batches per pass (the bf16 sweep of round 3 was flat; is this engine's?)
set -o pipefail
tag=${1:-r05m}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_fp8_pipelines_coalesce_sweep.txt
: > $out
for st in 2 3; do for co in 6 8 10 12 16; do
  timeout -k 10 120 python bench.py --precision fp8 --streams $st --coalesce $co --steps 1920 --warmup 10 --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg 2>/dev/null \
    | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $st coalesce $co', d['value'], d['ms_per_step'])" >> $out || exit 1
done; done
cat $out
