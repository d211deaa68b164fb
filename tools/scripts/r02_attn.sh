#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "attention or attn or parity or pipeline or edges" > gpurun_out/r02v_attn_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r02v_attn_tests.log
[ $rc -eq 0 ] || exit $rc
for steps in 20 200; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps $steps --warmup 5 2>/dev/null | cut -c1-170
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --stage-profile --streams 1 --coalesce 1 2>&1 >/dev/null | grep -v amdgpu.ids | head -8
timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --stage-profile --streams 1 --coalesce 10 2>&1 >/dev/null | grep -v amdgpu.ids | head -8
