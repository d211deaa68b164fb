#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "chain" > gpurun_out/r02w_chain_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r02w_chain_tests.log
[ $rc -eq 0 ] || exit $rc
for m in 8000; do
    echo "== M=$m"
    timeout -k 10 600 python tools/chain_stamps.py $m 2>&1 | grep "chain stamps" | tail -12
done | tee gpurun_out/r02w_chain_stamps2.txt
