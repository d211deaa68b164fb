"""Compare per-kernel durations between two rocprofv3 kernel traces (e.g. 1 pipeline vs 4 pipelines) and report the
concurrency of the second: time with 0/1/2/... kernels in flight, and the per-kernel slowdown under sharing.
    python tools/trace_overlap.py <trace_1pipe.csv> <trace_4pipe.csv>"""
import csv
import statistics
import sys
from collections import defaultdict


def load(f):
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((r["Kernel_Name"][:48] + "/" + r.get("Grid_Size_X", r.get("Grid_Size", "?")), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return rows


a, b = load(sys.argv[1]), load(sys.argv[2])


def med(rows):
    d = defaultdict(list)
    for n, s, e in rows:
        d[n].append((e - s) / 1000)
    return {n: (statistics.median(v), len(v), sum(v)) for n, v in d.items()}


ma, mb = med(a), med(b)
print(f"{'kernel/grid':58s} {'iso med':>8s} {'shared med':>10s} {'x':>5s} {'share of shared kernel-time':>8s}")
tot = sum(v[2] for v in mb.values())
for n, (m, c, s) in sorted(mb.items(), key=lambda kv: -kv[1][2])[:24]:
    ia = ma.get(n, (float('nan'),))[0]
    print(f"{n:58s} {ia:8.1f} {m:10.1f} {m / ia if ia == ia else 0:5.2f} {100 * s / tot:6.1f}%")
# concurrency histogram of b over its steady part (middle 60 %)
ev = []
t0, t1 = min(s for _, s, _ in b), max(e for _, _, e in b)
lo, hi = t0 + 0.3 * (t1 - t0), t0 + 0.9 * (t1 - t0)
for _, s, e in b:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
cur, last, hist = 0, ev[0][0], defaultdict(float)
for t, d in ev:
    if lo <= t <= hi:
        hist[cur] += t - max(last, lo)
    cur += d
    last = t
T = sum(hist.values())
print("kernels in flight: " + "  ".join(f"{k}: {100 * v / T:.1f}%" for k, v in sorted(hist.items())))
print(f"mean in flight {sum(k * v for k, v in hist.items()) / T:.2f}")
