#!/bin/bash
# A/B of two builds of the library: per-stage table (one pipeline, one batch per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp cassnat_asr_public_amd/libcassnat_hip.so /tmp/lib_keep.so
for v in ${AB_ORDER:-A B A B}; do
  cp ab/lib$v.so cassnat_asr_public_amd/libcassnat_hip.so
  echo "== $v"
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 10 --stage-profile --streams 1 --coalesce 1 2>&1 >/dev/null | grep -v amdgpu.ids | grep "${1:-conv2}"
done | tee gpurun_out/ab_stage.txt
cp /tmp/lib_keep.so cassnat_asr_public_amd/libcassnat_hip.so
