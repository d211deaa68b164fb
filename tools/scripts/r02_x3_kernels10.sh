#!/bin/bash
# kernel-trace stats of the bf16x3 engine at the benchmark's launch width (one pipeline, ten batches per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/x3k10
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x3k10 -o b -- python3 bench.py --precision bf16x3 --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 40 --warmup 10 --streams 1 --coalesce 10 > gpurun_out/x3k10.json 2>/dev/null || exit 1
f=$(find gpurun_out/x3k10 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/x3k10_stats.csv
rm -rf gpurun_out/x3k10
head -16 gpurun_out/x3k10_stats.csv | cut -c1-170
cut -c1-200 gpurun_out/x3k10.json
