#!/bin/bash
# decode pipelines per GPU (DecodePipelines)
for s in 2 3 4 5 6 8; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --streams $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelines', $s, d['value'], d['ms_per_step'])"
done
