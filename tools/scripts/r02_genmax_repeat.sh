#!/bin/bash
# per-dispatch durations of the generator tail launched back to back (tools/probes/genmax_repeat.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gmrep
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gmrep -o t -- python3 tools/probes/genmax_repeat.py > gpurun_out/gmrep.log 2>&1 || exit 1
f=$(find gpurun_out/gmrep -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/gmrep.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "genmax" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cur = None
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if r["Kernel_Name"] != cur:
        cur = r["Kernel_Name"]
        print("\n" + cur[:60], end=": ")
    print(f"{d:.1f}", end=" ")
print()
PY
