#!/bin/bash
# conv1's bordered image from the matrix-core kernel (bf16 and e4m3 engines) - tests, then A/B against the VALU kernels
# (bf16: CASSNAT_CONV1_MFMA=1 selects the matrix-core form; e4m3: CASSNAT_CONV1_F8_VALU=1 the VALU form;
# r05g was run when the bf16 switch was still CASSNAT_CONV1_VALU=1 with the matrix-core form as the default)
set -o pipefail
tag=${1:-r05f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_edges.py -x -q -m gpu -k "conv or parity or merged or agreement or bench or config" > gpurun_out/${tag}_tests.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_tests.log
[ $rc -ne 0 ] && exit $rc
out=gpurun_out/${tag}_conv1_mfma_ab.txt
: > $out
one() { local label=$1 prec=$2; shift 2
  env "$@" timeout -k 10 120 python bench.py --precision $prec --steps 2000 --warmup 10 --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg 2>/dev/null \
    | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['stage_ms'].get('conv1'), d['stage_ms'].get('conv2'))" >> $out
}
for rep in 1 2 3; do
one bf16_valu_conv1 bf16 X=1 &&
one bf16_mfma_conv1 bf16 CASSNAT_CONV1_MFMA=1 &&
one fp8_valu_conv1 fp8 CASSNAT_CONV1_F8_VALU=1 &&
one fp8_mfma_conv1 fp8 X=1 || exit 1
done
cat $out
