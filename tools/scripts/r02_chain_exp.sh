#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in "-DCH_EXP_NO_READ -DCH_EXP_NO_DMA -DCH_EXP_NO_BARRIER"; do
  for m in 8000; do
    echo "== flags: '$f' M=$m"
    timeout -k 10 600 python tools/chain_stamps.py $m $f 2>&1 | grep "chain stamps" | tail -12 | grep "S1\|S3\|S5\|total"
  done
done | tee gpurun_out/r02w_chain_exp5.txt
