#!/bin/bash
# split-bf16 engine after a kernel change: its unit tests, the parity gate, a short bench line, the stage profile and the FFN stamps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/scripts/r02_x3b.sh || exit 1
timeout -k 10 600 python tools/ffn_x3_stamps.py 6000 2048 > gpurun_out/r02q_stamps_6000.log 2>&1; tail -5 gpurun_out/r02q_stamps_6000.log
