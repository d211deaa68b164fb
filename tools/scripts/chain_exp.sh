cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dbg in 2; do
  CASSNAT_CHAIN_DEBUG=$dbg CASSNAT_CHAIN_REPEAT=20 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain_e$dbg -o chain -- python3 -m pytest tests/test_gpu_kernels.py -q -k "chain and 8000-2048-768" > gpurun_out/chain_exp$dbg.log 2>&1
  echo "EXP DBG=$dbg"; python3 tools/kernel_times.py gpurun_out/prof_chain_e$dbg chain
done
