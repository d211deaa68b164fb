#!/bin/bash
# whole GPU suite in one process, the side timings, then the round's measurement set
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -x -m gpu > gpurun_out/r02m_gpu_tests_all.log 2>&1; rc=$?
tail -4 gpurun_out/r02m_gpu_tests_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_esa.py --reps 12 --same-seed 2>/dev/null | tail -1 > gpurun_out/r02m_esa.json; cut -c1-500 gpurun_out/r02m_esa.json
