#!/bin/bash
# round 3, first GPU call: the whole GPU suite (new: merged ragged passes, area workspace, predicted rows, RCCL world-size-1),
# then a short bench in the default regime and through the forced-dist path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest ${R03_TESTS:-tests} -q -x -m gpu > gpurun_out/r03a_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r03a_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03a_bench20.json 2> gpurun_out/r03a_bench20.err; rc=$?
tail -3 gpurun_out/r03a_bench20.err; cut -c1-600 gpurun_out/r03a_bench20.json
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-parity-engine --no-uncoalesced > gpurun_out/r03a_bench_dist.json 2> gpurun_out/r03a_bench_dist.err; rc=$?
tail -3 gpurun_out/r03a_bench_dist.err; cut -c1-400 gpurun_out/r03a_bench_dist.json
exit $rc
