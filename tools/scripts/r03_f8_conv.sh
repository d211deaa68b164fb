#!/bin/bash
# e4m3 conv front-end: tests, config 5 accuracy, fp8 / bf16 engines side by side
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv" > gpurun_out/r03v_tests.log 2>&1 || { tail -40 gpurun_out/r03v_tests.log; exit 1; }
tail -3 gpurun_out/r03v_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "fp8 or config5" > gpurun_out/r03v_tests_fp8.log 2>&1 || { tail -30 gpurun_out/r03v_tests_fp8.log; exit 1; }
tail -3 gpurun_out/r03v_tests_fp8.log
for prec in fp8 bf16 fp8 bf16; do
  timeout -k 10 300 python bench.py --precision $prec --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 200 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_conv2']['avg_launch_us'], d['stage_ms'])" || exit 1
done | tee gpurun_out/r03v_bench_fp8_vs_bf16.txt
