#!/bin/bash
# whole GPU suite in one process + the side timings
timeout -k 10 1000 python -m pytest tests -q -x -m gpu > gpurun_out/full_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/full_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_conformer.py
