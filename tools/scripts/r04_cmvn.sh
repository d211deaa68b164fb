#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -m gpu -x -q -k "cmvn or cli" > gpurun_out/r04c_tests.log 2>&1 || { tail -40 gpurun_out/r04c_tests.log; exit 1; }
tail -2 gpurun_out/r04c_tests.log
timeout -k 10 300 python tools/ragged_cli_bench.py --utts 6000 2>gpurun_out/r04c_cli.err | tail -1 > gpurun_out/r04c_ragged_cli_6000utts_cmvn.json || { tail -20 gpurun_out/r04c_cli.err; exit 1; }
python3 -c "
import json
d=json.load(open('gpurun_out/r04c_ragged_cli_6000utts_cmvn.json'))
for k in ('plain','pipelined','pipelined_cmvn_in_dataset','pipelined_again','pipelined_4_loader_workers','preloaded'):
    print(k, d[k]['utt_per_s'], d[k]['seconds'], d[k].get('worker_host_seconds'))
print(d['result_files_identical'], d['global_cmvn'])
"
