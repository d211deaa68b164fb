#!/bin/bash
# A/B of two builds of the library (ab/libA.so, ab/libB.so) on one box, alternating: headline at $1 steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp cassnat_asr_public_amd/libcassnat_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do
  for v in A B; do
    cp ab/lib$v.so cassnat_asr_public_amd/libcassnat_hip.so
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps ${1:-200} --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" || exit 1
  done
done | tee gpurun_out/ab_bench.txt
cp /tmp/lib_keep.so cassnat_asr_public_amd/libcassnat_hip.so
