#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "chain or parity or pipeline or edges" > gpurun_out/r02w_chain_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r02w_chain_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/chain_stamps.py 8000 2>&1 | grep "chain stamps" | tail -12 | grep "S1\|S3\|S5\|total"
bash tools/scripts/ab_bench.sh 200
