#!/bin/bash
# Throughput of the fp8 engine by which products run in e4m3 (same box, 2000 timed steps each, twice).
set -o pipefail
tag=${1:-r05c}
out=gpurun_out/${tag}_fp8_scope_speed.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $out
one() { # label, precision, env...
  local label=$1 prec=$2; shift 2
  env "$@" timeout -k 10 120 python bench.py --precision $prec --steps 2000 --warmup 10 --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg 2>/dev/null \
    | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'])" >> $out
}
for rep in 1 2; do
one bf16 bf16 X=1 &&
one fp8_all fp8 X=1 &&
one fp8_conv2_ffn fp8 CASSNAT_NO_LINEAR_F8=1 &&
one fp8_conv2_ffn8to11 fp8 CASSNAT_NO_LINEAR_F8=1 CASSNAT_FP8_LAYERS=0xf00 &&
one fp8_conv2_ffn6to11 fp8 CASSNAT_NO_LINEAR_F8=1 CASSNAT_FP8_LAYERS=0xfc0 &&
one fp8_conv2 fp8 CASSNAT_NO_LINEAR_F8=1 CASSNAT_FP8_LAYERS=0 &&
one fp8_ffn fp8 CASSNAT_NO_CONV2_F8=1 || exit 1
done
cat $out
