#!/bin/bash
# is the slowdown of concurrent row-chain launches clock (same s_memtime ticks, longer wall time) or contention (more ticks)?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 1 3; do
  CASSNAT_CHAIN_STAMPS=1 CASSNAT_CHAIN_STREAMS=$n CASSNAT_CHAIN_REPEAT=81 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain_d$n -o chain -- python3 -m pytest tests/test_gpu_kernels.py -q -s -k "chain and 8000-2048-768 and True-3" > gpurun_out/chain_concd$n.log 2>&1
  echo "streams=$n"; python3 tools/kernel_times.py gpurun_out/prof_chain_d$n chain; grep -i "stamp" gpurun_out/chain_concd$n.log | tail -3
done
