#!/bin/bash
# whole GPU suite, then the headline at the driver's step count and at 200 (no extras), and the per-stage table
TAG=${1:-r02u}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/${TAG}_gpu_tests_all.log 2>&1; rc=$?
tail -3 gpurun_out/${TAG}_gpu_tests_all.log
[ $rc -eq 0 ] || exit $rc
for steps in 20 200; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps $steps --warmup 5 2>/dev/null | cut -c1-170
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --stage-profile --streams 1 --coalesce 1 2>&1 >/dev/null | grep -v amdgpu.ids | head -12
