#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -m gpu -x -q > gpurun_out/r03w_tests.log 2>&1 || { tail -40 gpurun_out/r03w_tests.log; exit 1; }
tail -3 gpurun_out/r03w_tests.log
bash tools/scripts/ab_bench.sh 200
# the ragged leg, A then B
for v in A B; do
  cp ab/lib$v.so cassnat_asr_public_amd/libcassnat_hip.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 40 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v ragged', d['ragged_set']['value'], d['ragged_set']['audio_seconds_per_second'], 'self_attn', d['stage_ms'].get('self_attention'))" || exit 1
done | tee gpurun_out/r03w_ragged_ab.txt
cp ab/libB.so cassnat_asr_public_amd/libcassnat_hip.so
