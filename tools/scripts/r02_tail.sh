#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 2>/dev/null | cut -c1-140; done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02s_tail -o t -- python3 bench.py --exit-after-timed --steps 20 --warmup 5 > gpurun_out/r02s_tail_bench.json 2>/dev/null || exit 1
cut -c1-150 gpurun_out/r02s_tail_bench.json
python3 tools/trace_tail.py gpurun_out/r02s_tail 19.5 15 | tee gpurun_out/r02s_tail.txt
rm -rf gpurun_out/r02s_tail
