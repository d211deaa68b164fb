#!/bin/bash
# round 3: software-pipelined activation in the row-chain kernel - parity tests, stamps, A/B against the previous kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edges.py tests/test_gpu_conformer.py -q -x -m gpu -k "chain or merged or bf16 or conf" > gpurun_out/r03b_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r03b_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/chain_stamps.py 80000 > gpurun_out/r03b_stamps_80000.txt 2>&1; tail -14 gpurun_out/r03b_stamps_80000.txt
bash tools/scripts/ab_bench.sh 200
timeout -k 10 600 python tools/ragged_cli_bench.py > gpurun_out/r03b_ragged_cli.json 2> gpurun_out/r03b_ragged_cli.err; rc=$?
tail -3 gpurun_out/r03b_ragged_cli.err; cat gpurun_out/r03b_ragged_cli.json | tail -1 | cut -c1-1500
exit $rc
