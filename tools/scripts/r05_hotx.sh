#!/bin/bash
# upper bound of a persistent row chain: bench with a build whose chain workgroups read L2-hot x rows (-DCH_EXP_HOT_X: wrong results)
set -o pipefail
tag=${1:-r05l}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_chain_hot_x_upper_bound.txt
: > $out
one() { local label=$1 prec=$2; shift 2
  env "$@" timeout -k 10 120 python bench.py --precision $prec --steps 2000 --warmup 10 --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg 2>/dev/null \
    | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['value'], d['ms_per_step'], d['stage_ms'].get('row_chain'))" >> $out
}
for rep in 1 2; do
one bf16 bf16 X=1 &&
one bf16_hot_x bf16 CASSNAT_HIP_LIB=$GRAFT_REPO_ROOT/ab/libhotx.so &&
one fp8 fp8 X=1 &&
one fp8_hot_x fp8 CASSNAT_HIP_LIB=$GRAFT_REPO_ROOT/ab/libhotx.so || exit 1
done
for blk in 300 450; do for lib in "" ab/libhotx.so; do echo "== block $blk lib $lib" >> $out; if [ -n "$lib" ]; then export CASSNAT_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset CASSNAT_HIP_LIB; fi
CASSNAT_CHAIN_STAMP_BLOCK=$blk CASSNAT_CHAIN_STAMPS=1 timeout -k 10 100 python tools/chain_stamps.py 80000 2>&1 | grep "chain stamps" | tail -11 >> $out; done; done
cat $out
