#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "chain" > gpurun_out/r04a_tests.log 2>&1 || { tail -30 gpurun_out/r04a_tests.log; exit 1; }
tail -2 gpurun_out/r04a_tests.log
for pf in 0 256; do
  echo "== CASSNAT_CHAIN_PREFETCH=$pf (80000 rows, bf16)"
  CASSNAT_CHAIN_PREFETCH=$pf CHAIN_X_MODE=19 timeout -k 10 200 python tools/chain_stamps.py 80000 2>&1 | grep "chain stamps" | tail -11
done | tee gpurun_out/r04a_chain_stamps_prefetch.txt
bash tools/scripts/ab_bench.sh 200
