#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "chain or parity or pipeline or edges" > gpurun_out/r02w_chain_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r02w_chain_tests.log
[ $rc -eq 0 ] || exit $rc
for f in "" "-DCH_EXP_PLAIN_STORE"; do
  for m in 8000 32768; do
    echo "== flags: '$f' M=$m"
    timeout -k 10 600 python tools/chain_stamps.py $m $f 2>&1 | grep "chain stamps" | tail -12 | grep "S5\|total\|landed\|S4"
  done
done | tee gpurun_out/r02w_chain_exp3.txt
