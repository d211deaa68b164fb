#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_conformer.py -q -x -s 2>&1 | grep -E "conformer|passed|failed"
timeout -k 10 300 python tools/time_conformer.py --streams 4 | cut -c150-420
timeout -k 10 300 python tools/time_conformer.py --transformer-encoder --streams 4 | cut -c150-420
