#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "generator" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -q -x -k "esa" -s 2>&1 | grep -E "ESA|passed|failed"
timeout -k 10 400 python tools/time_esa.py --samples 50 --reps 5
