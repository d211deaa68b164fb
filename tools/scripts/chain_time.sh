cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 180 python -m pytest tests/test_gpu_kernels.py -x -q -k chain > gpurun_out/chain_test.log 2>&1; tail -3 gpurun_out/chain_test.log
CASSNAT_CHAIN_REPEAT=20 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain_t -o chain -- python3 -m pytest tests/test_gpu_kernels.py -q -k "chain and 8000-2048-768 and True-3" > gpurun_out/chain_time.log 2>&1
python3 tools/kernel_times.py gpurun_out/prof_chain_t chain
CASSNAT_CHAIN_STAMPS=1 CASSNAT_CHAIN_REPEAT=5 timeout -k 10 120 python -m pytest tests/test_gpu_kernels.py -q -s -k "chain and 8000-2048-768 and True-3" 2>&1 | grep -E "stamps"
