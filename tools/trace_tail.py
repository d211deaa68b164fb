"""The last W ms of a rocprofv3 kernel trace (the timed region of a short bench run that ends the process): GPU busy share,
idle gaps over a threshold with the kernels on either side, and the kernel-time sum.
    python tools/trace_tail.py out [window_ms] [gap_us]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 18e6
    thr = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 20e3
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]))
    rows.sort()
    t_end = max(e for _, e, _ in rows)
    rows = [r for r in rows if r[1] >= t_end - win]
    t0 = rows[0][0]
    busy, cur_e, last_name, gaps = 0, None, None, []
    cur_s = None
    for s, e, n in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                if s - cur_e > thr:
                    gaps.append(((cur_e - t0) / 1e6, (s - cur_e) / 1e3, last_name, n))
            cur_s, cur_e, last_name = s, e, n
        elif e > cur_e:
            cur_e, last_name = e, n
    busy += cur_e - cur_s
    print("%d dispatches in the last %.2f ms: busy %.2f ms (%.1f %%)" % (len(rows), (t_end - t0) / 1e6, busy / 1e6, 100.0 * busy / (t_end - t0)))
    for at, g, a, b in gaps:
        print("  at %6.2f ms: idle %7.1f us  after %-40s before %s" % (at, g, a, b))


if __name__ == "__main__":
    main()
