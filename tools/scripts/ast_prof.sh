#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/time_ast.py
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ast -o e -- python3 tools/time_ast.py --reps 1 > gpurun_out/ast_time.json 2>/dev/null
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_ast/e_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "(2 runs)")
for r in rows[:18]: print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.2f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
