#!/usr/bin/env python3
"""GPU busy time against wall span over the tail of a rocprofv3 --kernel-trace CSV, and the largest idle gaps with the kernels either side.
usage: tools/trace_gaps.py <kernel_trace.csv> [from (default 0.5)] [to (default 0.8)]      (fractions of the kernel list)"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * f0):int(len(rows) * f1)]
short = lambda n: n.split("(")[0].replace("adt::", "").replace("void ", "")[:48]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy, cur_end, gaps = 0, t0, []
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        gaps.append((s - cur_end, short(rows[i - 1]["Kernel_Name"]) if i else "-", short(r["Kernel_Name"])))
        busy += e - s
    elif e > cur_end:
        busy += e - cur_end
    cur_end = max(cur_end, e)
print("kernels %d  span %.1f us  busy %.1f us (%.1f %%)" % (len(rows), (t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0)))
agg = Counter()
for g, a, b in gaps:
    agg[(a, b)] += g
print("idle by (kernel before -> kernel after), top 25:")
for (a, b), g in agg.most_common(25):
    n = sum(1 for x in gaps if x[1] == a and x[2] == b)
    print("  %9.1f us  x%-4d avg %6.1f   %s -> %s" % (g / 1e3, n, g / 1e3 / n, a, b))

nstep = max(1, sum(1 for r in rows if "k_adam" in r["Kernel_Name"]))
tot = Counter()
cnt = Counter()
for r in rows:
    tot[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[short(r["Kernel_Name"])] += 1
print("per step (%d steps in the window: span %.1f us per step), top 30 kernels:" % (nstep, (t1 - t0) / 1e3 / nstep))
for k, v in tot.most_common(30):
    print("  %8.1f us  %5.1f %%  x%-5.1f avg %7.1f  %s" % (v / 1e3 / nstep, 100.0 * v / sum(tot.values()), cnt[k] / nstep, v / 1e3 / cnt[k], k))
