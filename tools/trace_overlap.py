#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: kernel name, start and end relative to the step's first kernel, queue.
Shows which kernels of the backward ran under the chain kernels (side stream) and where the step has gaps.
usage: tools/trace_overlap.py <kernel_trace.csv> [n: the n-th step from the end (default 3); -n: the n-th step from the start]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_step_begin" in r["Kernel_Name"]]
print("steps in trace:", len(starts), " queues:", sorted({r.get("Queue_Id", "?") for r in rows}))
i0, i1 = (starts[-back - 1], starts[-back]) if back > 0 else (starts[-back], starts[-back + 1])
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("adt::", "").split("(")[0][:48]
    print("%-48s q%-3s %8.1f %8.1f  dur %6.1f  %s" % (name, r.get("Queue_Id", "?"), s / 1e3, e / 1e3, (e - s) / 1e3, "UNDER the previous kernels" if s < prev_end - 500 else ("gap %.1f" % ((s - prev_end) / 1e3) if s - prev_end > 1500 else "")))
    prev_end = max(prev_end, e)
print("step span %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
