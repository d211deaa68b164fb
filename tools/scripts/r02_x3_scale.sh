#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 2000 6000 12000 16384; do
  timeout -k 10 600 python tools/ffn_x3_stamps.py $m 2048 2>/dev/null | sed "s/^/M=$m  /"
done | tee gpurun_out/r02q_stamps_scale.txt
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/r02q_pmc_l2 -o p -- python3 tools/ffn_x3_stamps.py 12000 2048 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r02q_pmc_l2/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ffn_x3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, len(v), sum(v) / len(v))
PY
rm -rf gpurun_out/r02q_pmc_l2
