#!/bin/bash
# S5 (projection tail) with and without its stores
for st in 1 2; do
CASSNAT_CHAIN_STAMPS=$st CASSNAT_CHAIN_REPEAT=5 timeout -k 10 120 python -m pytest tests/test_gpu_kernels.py -q -s -k "chain and 8000-2048-768 and True-3" 2>&1 | grep -E "stamps\] (S3|S5|total|LNn|S4)"
done
