#!/usr/bin/env python3
"""Idle time between the kernels of the captured training step, from a rocprofv3 kernel trace (train_kernel_trace.csv):
the last replay's kernels sorted by start time; busy = union of the kernel intervals, idle = the rest of the replay's span,
listed by the kernel that FOLLOWS each gap.  Usage: train_gaps.py <kernel_trace.csv> <launches per step>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
ev = ev[-per:]
span = ev[-1][1] - ev[0][0]
busy, gaps, end = 0, [], ev[0][0]
for s, e, n in ev:
    if s > end:
        gaps.append((s - end, n))
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
print("span %.2f ms, busy %.2f ms, idle %.2f ms in %d gaps (median %.2f us, mean %.2f us)" % (
    span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3,
    sum(g for g, _ in gaps) / max(1, len(gaps)) / 1e3))
by = defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[n.split("(")[0][-60:]][0] += g
    by[n.split("(")[0][-60:]][1] += 1
for n, (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us idle before %3d x %s (%.1f us each)" % (g / 1e3, c, n, g / c / 1e3))
hist = defaultdict(int)
for g, _ in gaps:
    hist[min(int(g / 1000), 20)] += 1
print("gap histogram (us: count):", dict(sorted(hist.items())))
