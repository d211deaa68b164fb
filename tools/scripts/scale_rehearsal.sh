#!/bin/bash
# N = 2 rehearsal of bench.py's multi-rank path on ONE GPU (both ranks share it): RCCL cannot run two ranks on one device,
# so the collectives go over gloo; checks the launch contract, the weight broadcast and the per-step all-gather.
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 4 --backend gloo --no-cpu-baseline 2> gpurun_out/scale2.err | tail -1 | cut -c1-400
tail -3 gpurun_out/scale2.err
timeout -k 10 300 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-200 gpurun_out/bench_default.json
python3 -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['cpu_baseline'])"
