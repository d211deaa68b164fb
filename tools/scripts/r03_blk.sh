#!/bin/bash
# round 3: blocked Q|K|V / K|V outputs of the row-chain kernel's tail, read as such by the attention kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edges.py tests/test_gpu_pipeline.py tests/test_gpu_conformer.py -q -x -m gpu > gpurun_out/r03c_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r03c_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/chain_stamps.py 80000 > gpurun_out/r03c_stamps_80000.txt 2>&1; tail -12 gpurun_out/r03c_stamps_80000.txt
bash tools/scripts/ab_bench.sh 200
timeout -k 10 600 python tools/ragged_cli_bench.py > gpurun_out/r03c_ragged_cli.json 2> gpurun_out/r03c_ragged_cli.err; rc=$?
tail -3 gpurun_out/r03c_ragged_cli.err; cat gpurun_out/r03c_ragged_cli.json | tail -1 | cut -c1-1500
exit $rc
