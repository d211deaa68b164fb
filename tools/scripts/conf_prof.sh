#!/bin/bash
# conformer variants: timing + per-kernel profile of the shipped combination (transformer encoder, conformer decoder)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/time_conformer.py --streams 4 | cut -c1-400
timeout -k 10 300 python tools/time_conformer.py --transformer-encoder --streams 4 | cut -c1-400
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_conf -o e -- python3 tools/time_conformer.py --transformer-encoder --reps 3 > gpurun_out/conf_time.json 2>gpurun_out/conf_time.err
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_conf/e_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "(4 runs)")
for r in rows[:16]: print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
