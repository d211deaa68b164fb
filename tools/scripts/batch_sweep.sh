#!/bin/bash
# same total work, different packing: utterances per call x decode pipelines
for cfg in "32 4" "64 2" "64 4" "128 1" "128 2" "256 1"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --no-cpu-baseline --batch $1 --streams $2 --steps 24 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('batch', $1, 'streams', $2, d['value'], 'utt/s', d['ms_per_step'], 'ms/step', 'chain frac', r['frac'], 'avg us', r['avg_launch_us'])"
done
