#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "chain" 2>&1 | tail -2
timeout -k 10 600 python -m pytest tests/test_gpu_conformer.py -q -x -s 2>&1 | grep -E "conformer|passed|failed"
timeout -k 10 300 python tools/time_conformer.py
CASSNAT_NO_CHAIN=1 timeout -k 10 300 python tools/time_conformer.py | cut -c1-260
