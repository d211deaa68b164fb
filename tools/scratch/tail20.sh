#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg --steps 20 --warmup 5 --exit-after-timed"
for v in "" "--streams 1 --coalesce 20" "--plan 5,5,5,5"; do
  tag=$(echo "$v" | tr -c 'a-z0-9\n' '_')
  rm -rf gpurun_out/tl_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$tag -o t -- python3 bench.py $B $v 2>/dev/null | tail -1
  echo "== $v"; python3 tools/trace_tail.py gpurun_out/tl_$tag 15.5 8
  rm -rf gpurun_out/tl_$tag
done > gpurun_out/r04t_tail20.txt 2>&1
tail -80 gpurun_out/r04t_tail20.txt
