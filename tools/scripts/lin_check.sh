#!/bin/bash
# linear_out on the LDS-DMA tile kernel: parity, then bench with and without it
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or conv" > gpurun_out/lin_tests.log 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_conformer.py -q -x >> gpurun_out/lin_tests.log 2>&1
tail -3 gpurun_out/lin_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/lin_bench_on.json 2> gpurun_out/lin_bench_on.err
CASSNAT_NO_LINEAR_DMA=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/lin_bench_off.json 2> gpurun_out/lin_bench_off.err
cat gpurun_out/lin_bench_on.json gpurun_out/lin_bench_off.json | cut -c1-220
