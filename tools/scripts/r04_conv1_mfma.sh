#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -m gpu -x -q -k "fp8 or config5 or conv" > gpurun_out/r04j_tests.log 2>&1 || { tail -40 gpurun_out/r04j_tests.log; exit 1; }
tail -4 gpurun_out/r04j_tests.log
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export CASSNAT_CONV1_F8_VALU=1; else unset CASSNAT_CONV1_F8_VALU; fi
  timeout -k 10 300 python bench.py --precision fp8 --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 200 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('valu_conv1=$v', d['value'], d['ms_per_step'], d['stage_ms'].get('conv1'))" || exit 1
done | tee gpurun_out/r04j_conv1_mfma_ab.txt
