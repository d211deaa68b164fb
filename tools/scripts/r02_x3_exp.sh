#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in "" "-DFX_EXP_NO_WLOAD" "-DFX_EXP_NO_XREAD" "-DFX_EXP_NO_WLOAD -DFX_EXP_NO_XREAD"; do
  echo "== flags: $f"
  timeout -k 10 600 python tools/ffn_x3_stamps.py 6000 2048 $f 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r02q_x3_exp.txt
