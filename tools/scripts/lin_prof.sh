#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lin -o b -- python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/bench_prof_lin.json 2>/dev/null
python3 tools/kernel_times.py gpurun_out/prof_lin conv2 gemm chain attn > gpurun_out/lin_ktimes.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lin1 -o b -- python3 bench.py --no-cpu-baseline --steps 20 --streams 1 > gpurun_out/bench_prof_lin1.json 2>/dev/null
python3 tools/kernel_times.py gpurun_out/prof_lin1 conv2 gemm chain attn > gpurun_out/lin_ktimes1.txt
cat gpurun_out/lin_ktimes1.txt
