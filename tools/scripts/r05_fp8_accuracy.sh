#!/bin/bash
# Where the fp8 engine's arg-max flips come from: tools/fp8_accuracy.py under the library's build switches.
#   bash tools/scripts/r05_fp8_accuracy.sh <tag> [batch]
set -o pipefail
tag=${1:-r05a}
B=${2:-48}
out=gpurun_out/${tag}_fp8_accuracy.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out
run() { env "$@" timeout -k 10 200 python tools/fp8_accuracy.py --engines fp8 --batch $B 2>/dev/null | grep "argmax" >> $out; }
timeout -k 10 200 python tools/fp8_accuracy.py --engines bf16,fp8 --batch $B 2>/dev/null | grep "argmax" >> $out &&
run CASSNAT_NO_LINEAR_F8=1 &&
run CASSNAT_NO_LINEAR_F8=1 CASSNAT_FP8_LAYERS=0xfc0 &&
run CASSNAT_NO_LINEAR_F8=1 CASSNAT_FP8_LAYERS=0xf00 &&
run CASSNAT_NO_LINEAR_F8=1 CASSNAT_FP8_LAYERS=0 &&
run CASSNAT_FP8_LAYERS=0 &&
run CASSNAT_NO_CONV2_F8=1 &&
run CASSNAT_NO_CONV2_F8=1 CASSNAT_FP8_LAYERS=0xff0 &&
run CASSNAT_NO_CONV2_F8=1 CASSNAT_FP8_LAYERS=0xfc0 &&
run CASSNAT_NO_CONV2_F8=1 CASSNAT_FP8_LAYERS=0x03f
cat $out
