#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03b_x3_trace -o t -- python3 bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 60 --streams 1 --coalesce 10 > gpurun_out/r03b_x3_trace_bench.json 2>/dev/null || exit 1
cut -c1-160 gpurun_out/r03b_x3_trace_bench.json
python3 tools/trace_busy.py gpurun_out/r03b_x3_trace 16 | tee gpurun_out/r03b_x3_trace_busy.txt
rm -rf gpurun_out/r03b_x3_trace
timeout -k 10 300 python bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --stage-profile --streams 1 --coalesce 1 2>&1 >/dev/null | grep -v amdgpu.ids | head -16
