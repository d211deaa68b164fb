#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ast.py tests/test_gpu_ctcbeam.py -q -x -m gpu > gpurun_out/r03e_ast.log 2>&1; rc=$?
tail -25 gpurun_out/r03e_ast.log
[ $rc -eq 0 ] || exit $rc
for p in bf16x3 fp32 bf16; do timeout -k 10 300 python tools/time_ast.py --precision $p 2>&1 | tail -2; done | tee gpurun_out/r03e_ast_time.txt
