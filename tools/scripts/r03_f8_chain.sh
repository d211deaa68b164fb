#!/bin/bash
# e4m3 feed-forward form of the chain kernel: tests, phase stamps of both forms, fp8 / bf16 engines side by side
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chain" > gpurun_out/r03t_tests.log 2>&1 || { tail -30 gpurun_out/r03t_tests.log; exit 1; }
tail -3 gpurun_out/r03t_tests.log
for mode in 19 51; do
  echo "== CHAIN_X_MODE=$mode (80000 rows)"
  CHAIN_X_MODE=$mode timeout -k 10 200 python tools/chain_stamps.py 80000 2>&1 | grep "chain stamps" | tail -11
done | tee gpurun_out/r03t_chain_stamps_bf16_vs_f8.txt
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "fp8 or config5" > gpurun_out/r03t_tests_fp8.log 2>&1 || { tail -30 gpurun_out/r03t_tests_fp8.log; exit 1; }
tail -3 gpurun_out/r03t_tests_fp8.log
for prec in fp8 bf16; do
  timeout -k 10 300 python bench.py --precision $prec --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 200 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec', d['value'], d['ms_per_step'], d['roofline']['frac'])" || exit 1
done | tee gpurun_out/r03t_bench_fp8_vs_bf16.txt
