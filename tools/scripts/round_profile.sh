# Round-end measurement set (run on the GPU box through gpurun): default bench line, kernel-trace stats at the default
# 4 pipelines and at 1 pipeline, and the two PMC passes the roofline 'traffic' figures come from.  Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
set -x
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -o b -- python3 bench.py --no-cpu-baseline --steps 20 > gpurun_out/bench_prof_default.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_1pipe -o b -- python3 bench.py --no-cpu-baseline --steps 20 --streams 1 > gpurun_out/bench_prof_1pipe.json 2>/dev/null
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --streams 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --streams 1 > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out
tail -2 gpurun_out/bench_default.err; cut -c1-300 gpurun_out/bench_default.json
