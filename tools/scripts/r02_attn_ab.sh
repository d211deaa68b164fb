#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "attention or attn or parity or pipeline or edges or ast or esa" > gpurun_out/r02v_attn_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r02v_attn_tests.log
[ $rc -eq 0 ] || exit $rc
AB_ORDER="A B A B" bash tools/scripts/ab_stage.sh "attention"
bash tools/scripts/ab_bench.sh 200
