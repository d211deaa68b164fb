#!/bin/bash
# ESA (sample_num 50): samples per decoder-side pass against time per batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for prec in bf16 bf16x3; do
for g in 16 10 13 14 17 25 28 50; do
  timeout -k 10 200 python tools/time_esa.py --precision $prec --group $g --reps 3 --same-seed 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec group $g', d['sec_per_batch'], d['utt_per_sec'], d['tokens_max'], d['all_runs_sec'])" || echo "$prec group $g failed"
done
done | tee gpurun_out/r04r_esa_group_sweep.txt
