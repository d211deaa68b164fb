#!/bin/bash
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_pipeline.py -q -x -k "pipelines or cli" 2>&1 | tail -4
timeout -k 10 300 python bench.py --no-cpu-baseline | cut -c1-220
timeout -k 10 300 python bench.py --no-cpu-baseline --streams 1 --steps 40 | cut -c1-220
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 40 --warmup 4 --backend gloo --no-cpu-baseline 2> gpurun_out/scale2.err | tail -1 | cut -c1-220
