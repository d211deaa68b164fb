#!/bin/bash
# Round-2 measurement set (GPU box, through gpurun): default bench line, kernel-trace stats at the default pipelines and at one
# pipeline with one batch per pass, the PMC passes the roofline 'traffic' figures come from (separate passes, --pmc with
# --kernel-trace only), and the SQ MFMA-utilisation pass.  Outputs under gpurun_out/ (copied into profiles/ afterwards).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_steps20.json 2>/dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof_default -o b -- python3 bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 40 > gpurun_out/${TAG}_bench_prof_default.json 2>/dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof_1pipe -o b -- python3 bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 5 --streams 1 --coalesce 1 > gpurun_out/${TAG}_bench_prof_1pipe.json 2>/dev/null || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 10 --streams 1 --coalesce 10 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_write -o p -- python3 bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 10 --streams 1 --coalesce 10 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_mfma -o p -- python3 bench.py --no-cpu-baseline --no-parity-engine --no-uncoalesced --steps 20 --warmup 10 --streams 1 --coalesce 10 > /dev/null 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out gpurun_out/${TAG}_pmc_mfma 10
for d in default 1pipe; do f=$(find gpurun_out/${TAG}_prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_kernel_stats_$d.csv; done
tail -2 gpurun_out/${TAG}_bench_default.err; cut -c1-260 gpurun_out/${TAG}_bench_default.json; cut -c1-200 gpurun_out/${TAG}_bench_steps20.json
head -12 gpurun_out/${TAG}_kernel_stats_default.csv | cut -c1-150
