"""Find what a stall is made of in a rocprofv3 kernel trace: kernels longer than a threshold and idle gaps between consecutive
kernels (same device) longer than it.
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 tools/esa_phases.py
    python tools/trace_gaps.py out [threshold_ms]"""
import csv
import glob
import os
import sys


def main():
    path, thr = sys.argv[1], float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:80]))
    rows.sort()
    print(len(rows), "dispatches")
    last_end, last_name = None, None
    for s, e, n in rows:
        if e - s > thr:
            print(f"LONG KERNEL {(e - s) / 1e6:8.2f} ms  {n}")
        if last_end is not None and s - last_end > thr:
            print(f"GAP        {(s - last_end) / 1e6:8.2f} ms  after {last_name}  before {n}")
        if last_end is None or e > last_end:
            last_end, last_name = e, n


if __name__ == "__main__":
    main()
