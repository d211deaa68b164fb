#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02h_esa_trace -o t -- python3 tools/esa_phases.py > gpurun_out/r02h_esa_phases.txt 2>&1
tail -9 gpurun_out/r02h_esa_phases.txt
python3 tools/trace_gaps.py gpurun_out/r02h_esa_trace 5 | tail -40
rm -rf gpurun_out/r02h_esa_trace
