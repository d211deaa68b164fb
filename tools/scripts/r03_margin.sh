#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity-engine"
for s in 20 200; do timeout -k 10 300 python bench.py $B --steps $s --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps $s:', d['value'], d['ms_per_step'], 'chain frac', r['frac'], 'alone', r['isolated_at_width_frac'], 'missed', d['config']['row_predictions_missed'], '1bpp', d['one_batch_per_pass']['value'], 'ragged', d['ragged_set'] and (d['ragged_set']['value'], d['ragged_set']['audio_seconds_per_second'], d['ragged_set']['row_predictions_missed']))"; done
timeout -k 10 600 python tools/ragged_cli_bench.py --utts 6000 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v['utt_per_s'], v.get('row_predictions_missed')) for k,v in d.items() if isinstance(v,dict) and 'utt_per_s' in v})"
timeout -k 10 300 python -m pytest tests/test_gpu_edges.py -q -x -m gpu -k "merged or pipelines" 2>&1 | tail -2
