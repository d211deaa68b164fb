#!/bin/bash
# how do concurrent row-chain launches share the chip?  80 launches round-robin over n streams
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 1 2 3 4 6; do
  CASSNAT_CHAIN_STREAMS=$n CASSNAT_CHAIN_REPEAT=81 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_chain_c$n -o chain -- python3 -m pytest tests/test_gpu_kernels.py -q -k "chain and 8000-2048-768 and True-3" > gpurun_out/chain_conc$n.log 2>&1
  echo "streams=$n"; python3 tools/kernel_times.py gpurun_out/prof_chain_c$n chain
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_chain_c$n/**/*kernel_trace.csv",recursive=True)[0]
r=[(int(x["Start_Timestamp"]),int(x["End_Timestamp"])) for x in csv.DictReader(open(f)) if "chain" in x["Kernel_Name"]]
r.sort(); r=r[1:]
print("  launches",len(r),"span us",(max(e for s,e in r)-r[0][0])/1000,"-> per launch",(max(e for s,e in r)-r[0][0])/1000/len(r))
PY
done
