#!/bin/bash
# decode pipelines x hardware queues
for q in 4 8 16; do
for s in 2 4 6 8 12; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --no-cpu-baseline --streams $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('hwq', $q, 'streams', $s, d['value'], d['ms_per_step'])"
done
done
