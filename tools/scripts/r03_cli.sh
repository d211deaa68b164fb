#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/ragged_cli_bench.py > gpurun_out/r03i_ragged_cli.json 2> gpurun_out/r03i_ragged_cli.err; rc=$?
tail -3 gpurun_out/r03i_ragged_cli.err; tail -1 gpurun_out/r03i_ragged_cli.json | cut -c1-1400
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/ragged_cli_bench.py --utts 6000 > gpurun_out/r03i_ragged_cli_6000.json 2> gpurun_out/r03i_ragged_cli_6000.err; rc=$?
tail -1 gpurun_out/r03i_ragged_cli_6000.json | cut -c1-1400
exit $rc
