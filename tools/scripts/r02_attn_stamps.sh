#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 32 320; do echo "== B=$b"; CASSNAT_ATTN_STAMPS=1 timeout -k 10 120 python tools/attn_stamps.py $b 2>&1 | grep -v amdgpu.ids | tail -8; done | tee gpurun_out/r02v_attn_stamps.txt
