#!/bin/bash
# Round artifacts at HEAD (GPU box, through gpurun):  bash tools/scripts/round_artifacts.sh <tag>
# the whole -m gpu suite in one process, the default bench line at the driver's 20 steps (with its steady_state, parity, fp8 legs).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-round}
timeout -k 10 1100 python -m pytest tests -m gpu -q -x ${GPU_TESTS:-} > gpurun_out/${T}_gpu_tests_all.log 2>&1; rc=$?
tail -4 gpurun_out/${T}_gpu_tests_all.log
[ $rc -ne 0 ] && exit $rc
[ "${2:-bench}" = "bench" ] || exit 0
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_steps20.json 2> gpurun_out/${T}_bench_steps20.err || exit 1
python3 -c "import json; d=json.load(open('gpurun_out/${T}_bench_steps20.json')); print('bf16 20 steps', d['value'], 'steady', d['steady_state']['value'], 'frac', d['roofline']['frac'], 'x3', d['parity_engine']['value'], d['parity_engine'].get('hyp_agreement'), 'fp8', d['fp8_engine']['value'])"
