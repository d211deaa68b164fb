#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k attention 2>&1 | tail -3
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn 2wg/cu', d['value'], d['ms_per_step'], d['stage_ms'].get('self_attention'))" || exit 1
done
