#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/ffn_x3_stamps.py 2000 2048 > gpurun_out/r02q_stamps_2000.log 2>&1; rc=$?; tail -8 gpurun_out/r02q_stamps_2000.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ffn_x3_stamps.py 6000 2048 > gpurun_out/r02q_stamps_6000.log 2>&1; tail -5 gpurun_out/r02q_stamps_6000.log
