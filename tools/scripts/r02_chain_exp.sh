#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 8000 32768 80000; do
    echo "== M=$m"
    timeout -k 10 600 python tools/chain_stamps.py $m 2>&1 | grep "chain stamps" | tail -12
done | tee gpurun_out/r02w_chain_stamps.txt
