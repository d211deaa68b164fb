"""Per-kernel launch durations from a rocprofv3 --kernel-trace CSV: for every (kernel name containing one of the given
substrings, grid size) print count, min, median and max in microseconds.
    python tools/kernel_times.py <dir-or-csv> chain ffn_fused"""
import csv
import glob
import os
import statistics
import sys
from collections import defaultdict

path = sys.argv[1]
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in sys.argv[2:]):
            acc[(n[:56], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for (n, g), v in sorted(acc.items()):
    print(f"{n:56s} grid {g:>7s} n={len(v):4d} min {min(v):8.2f} med {statistics.median(v):8.2f} max {max(v):8.2f} us")
