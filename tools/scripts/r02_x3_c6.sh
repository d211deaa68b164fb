#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02q_x3_trace6 -o t -- python3 bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 72 --warmup 6 --streams 1 --coalesce 6 > gpurun_out/r02q_x3_trace6_bench.json 2>/dev/null || exit 1
cut -c1-160 gpurun_out/r02q_x3_trace6_bench.json
python3 tools/trace_busy.py gpurun_out/r02q_x3_trace6 | tee gpurun_out/r02q_x3_trace6_busy.txt
rm -rf gpurun_out/r02q_x3_trace6
