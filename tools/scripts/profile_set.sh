#!/bin/bash
# Measurement set of one engine (GPU box, through gpurun):  bash tools/scripts/profile_set.sh <tag> [bf16|bf16x3|fp8]
# kernel-trace stats in the default regime (2 pipelines x 10 batches per pass) and with one pipeline, then the PMC passes behind the
# roofline `traffic` figures (separate passes, --pmc with nothing but --kernel-trace, the program directly after --, bench.py
# --exit-after-timed so that every launch belongs to a pass of the timed width) and the SQ MFMA-utilisation pass.
# Outputs under gpurun_out/<tag>_* (copy what is to be judged into profiles/).
TAG=${1:-prof}; PREC=${2:-bf16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--precision $PREC --no-cpu-baseline --no-parity-engine --no-uncoalesced --no-ragged-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof_default -o b -- python3 bench.py $B --steps 40 --exit-after-timed > gpurun_out/${TAG}_bench_prof_default.json 2>/dev/null || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof_1x10 -o b -- python3 bench.py $B --steps 40 --streams 1 --coalesce 10 --exit-after-timed > gpurun_out/${TAG}_bench_prof_1x10.json 2>/dev/null || exit 1
for d in default 1x10; do f=$(find gpurun_out/${TAG}_prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_kernel_stats_$d.csv; done
python3 tools/kernel_times.py gpurun_out/${TAG}_prof_1x10 chain attention conv2 conv1 genmax ffn proj > gpurun_out/${TAG}_kernel_times_by_grid_1x10.txt
if [ "${3:-pmc}" = "pmc" ]; then
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_fetch -o p -- python3 bench.py $B --steps 20 --warmup 10 --streams 1 --coalesce 10 --exit-after-timed > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${TAG}_pmc_write -o p -- python3 bench.py $B --steps 20 --warmup 10 --streams 1 --coalesce 10 --exit-after-timed > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_mfma -o p -- python3 bench.py $B --steps 20 --warmup 10 --streams 1 --coalesce 10 --exit-after-timed > /dev/null 2>&1 || exit 1
  mkdir -p gpurun_out/${TAG}_pmc && python3 tools/pmc_summary.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc_mfma 10
fi
rm -rf gpurun_out/${TAG}_prof_default gpurun_out/${TAG}_prof_1x10 gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_mfma
cat gpurun_out/${TAG}_bench_prof_default.json gpurun_out/${TAG}_bench_prof_1x10.json
head -14 gpurun_out/${TAG}_kernel_stats_default.csv | cut -c1-170
head -14 gpurun_out/${TAG}_kernel_stats_1x10.csv | cut -c1-170
