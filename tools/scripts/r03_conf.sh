#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_conformer.py tests/test_gpu_bf16x3.py -q -x -m gpu > gpurun_out/r03f_conf.log 2>&1; rc=$?
tail -25 gpurun_out/r03f_conf.log
exit $rc
