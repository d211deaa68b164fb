#!/bin/bash
# round artifacts at HEAD: whole GPU suite, default bench, fp8 / force-dist bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r03u}
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${T}_gpu_tests_all.log 2>&1; rc=$?
tail -4 gpurun_out/${T}_gpu_tests_all.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_steps20.json 2> gpurun_out/${T}_bench_steps20.err || exit 1
python3 -c "import json; d=json.load(open('gpurun_out/${T}_bench_steps20.json')); print('bf16 20 steps', d['value'], 'steady', d['steady_state']['value'], 'frac', d['roofline']['frac'], 'x3', d['parity_engine']['value'])"
timeout -k 10 300 python bench.py --precision fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-parity-engine > gpurun_out/${T}_bench_fp8_steps20.json 2>/dev/null || exit 1
python3 -c "import json; d=json.load(open('gpurun_out/${T}_bench_fp8_steps20.json')); print('fp8 20 steps', d['value'], 'steady', d.get('steady_state',{}).get('value'), 'frac', d['roofline']['frac'])"
