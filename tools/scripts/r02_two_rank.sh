#!/bin/bash
# rehearsal of the driver's N > 1 command on ONE GPU: two ranks over gloo (weights broadcast, per-batch all-gather of records)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo > gpurun_out/r02z_two_rank.json 2> gpurun_out/r02z_two_rank.err; rc=$?
tail -5 gpurun_out/r02z_two_rank.err | cut -c1-200
cut -c1-400 gpurun_out/r02z_two_rank.json
exit $rc
