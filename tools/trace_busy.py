"""How busy the GPU is in a rocprofv3 kernel trace, and with what: over the steady part of the run (the middle 60 % of the
dispatches by start time) prints the wall span, the time at least one kernel was running, the mean number of kernels in
flight while busy, and per kernel the launches, mean duration and share of the summed kernel time.
    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py ...
    python tools/trace_busy.py out"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
    rows.sort()
    n = len(rows)
    rows = rows[n // 5: n - n // 5]
    t0, t1 = rows[0][0], max(e for _, e, _ in rows)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    total = sum(e - s for s, e, _ in rows)
    print("%d dispatches over %.2f ms: GPU busy %.2f ms (%.1f %%), kernel time summed %.2f ms = %.2f kernels in flight while busy"
          % (len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), total / 1e6, total / busy))
    per = defaultdict(list)
    for s, e, k in rows:
        per[k].append(e - s)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
        print("%-62s n=%5d  mean %8.1f us  share %5.1f %%" % (k, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / total))


if __name__ == "__main__":
    main()
