#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edges.py tests/test_gpu_conformer.py tests/test_gpu_bf16x3.py tests/test_gpu_pipeline.py -q -x -m gpu > gpurun_out/r03g_tests.log 2>&1; rc=$?
tail -12 gpurun_out/r03g_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/chain_stamps.py 80000 > gpurun_out/r03g_stamps_80000.txt 2>&1; tail -12 gpurun_out/r03g_stamps_80000.txt
bash tools/scripts/ab_bench.sh 200
