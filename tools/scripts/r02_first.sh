#!/bin/bash
# round 2, first GPU call: the whole GPU suite, then the bench at the driver's step count and at the default one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/r02a_gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r02a_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a_bench_steps20.json 2> gpurun_out/r02a_bench_steps20.err || exit 1
cut -c1-400 gpurun_out/r02a_bench_steps20.json
timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine > gpurun_out/r02a_bench_steps200.json 2> gpurun_out/r02a_bench_steps200.err || exit 1
cut -c1-400 gpurun_out/r02a_bench_steps200.json
