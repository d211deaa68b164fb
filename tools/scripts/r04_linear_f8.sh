#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -m gpu -x -q -k "fp8 or config5 or conv or linear" > gpurun_out/r04k_tests.log 2>&1 || { tail -40 gpurun_out/r04k_tests.log; exit 1; }
tail -4 gpurun_out/r04k_tests.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export CASSNAT_NO_LINEAR_F8=1; else unset CASSNAT_NO_LINEAR_F8; fi
  timeout -k 10 300 python bench.py --precision fp8 --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 200 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16_linear_out=$v', d['value'], d['ms_per_step'], d['stage_ms'].get('linear_out_embed'), d['stage_ms'].get('conv2'))" || exit 1
done | tee gpurun_out/r04k_linear_f8_ab.txt
