#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_bf16x3.py -q -x -m gpu > gpurun_out/r02e_x3_kernels.log 2>&1; rc=$?
tail -15 gpurun_out/r02e_x3_kernels.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_pipeline.py -q -x -m gpu -k "parity_gate and bf16x3" > gpurun_out/r02e_x3_gate.log 2>&1; rc=$?
tail -12 gpurun_out/r02e_x3_gate.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 40 > gpurun_out/r02e_bench_x3.json 2> gpurun_out/r02e_bench_x3.err; tail -3 gpurun_out/r02e_bench_x3.err; cut -c1-200 gpurun_out/r02e_bench_x3.json
timeout -k 10 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 40 --stage-profile --streams 1 --coalesce 1 2>&1 >/dev/null | tail -25
