#!/bin/bash
set -e
timeout -k 10 800 python -m pytest tests/test_gpu_ast.py -q -x -s 2>&1 | grep -E "AST|passed|failed|Error|error" | tail -12
timeout -k 10 300 python tools/time_ast.py --streams 4
CASSNAT_NO_CHAIN=1 timeout -k 10 300 python tools/time_ast.py
