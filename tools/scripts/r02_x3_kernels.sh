#!/bin/bash
# kernel-trace stats of the bf16x3 engine (one pipeline, one batch per pass), with and without the projection kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in new gemm; do
  rm -rf gpurun_out/x3k_$v
  if [ $v = gemm ]; then export CASSNAT_NO_PROJ_X3=1; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x3k_$v -o b -- python3 bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 --streams 1 --coalesce 1 > gpurun_out/x3k_$v.json 2>/dev/null || exit 1
  f=$(find gpurun_out/x3k_$v -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/x3k_stats_$v.csv
  rm -rf gpurun_out/x3k_$v
  echo "== $v"; head -14 gpurun_out/x3k_stats_$v.csv | cut -c1-150
done
