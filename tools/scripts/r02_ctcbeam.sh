#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_ctcbeam.py -q -x -m gpu > gpurun_out/r02d_ctcbeam.log 2>&1; rc=$?
tail -40 gpurun_out/r02d_ctcbeam.log
exit $rc
