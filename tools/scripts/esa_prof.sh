#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_esa -o e -- python3 tools/time_esa.py --samples 50 --reps 2 > gpurun_out/esa_time.json 2>/dev/null
cat gpurun_out/esa_time.json
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_esa/e_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:14]: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
