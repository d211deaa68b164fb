#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rot in 1 0 1 0; do
  CASSNAT_FFN_X3_ROTATE=$rot timeout -k 10 200 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 60 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rotate', $rot, d['value'], d['ms_per_step'], d['stage_ms'].get('ffn_fused_x3'))" || exit 1
done 2>&1 | tee gpurun_out/r02i_x3_rot.txt
echo "--- ESA timing, default vs HSA_ENABLE_INTERRUPT=0"
timeout -k 10 300 python tools/time_esa.py --reps 12 --same-seed 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default   ', d['all_runs_sec'])"
HSA_ENABLE_INTERRUPT=0 timeout -k 10 300 python tools/time_esa.py --reps 12 --same-seed 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-interrupt', d['all_runs_sec'])"
echo "--- bench bf16, default vs HSA_ENABLE_INTERRUPT=0"
for rep in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default steps20', d['value'])"
HSA_ENABLE_INTERRUPT=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-interrupt steps20', d['value'])"
done
HSA_ENABLE_INTERRUPT=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 200 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-interrupt steps200', d['value'])"
