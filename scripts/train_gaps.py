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

# concurrency profile of the same replay: time with k kernels running, and which kernels run ALONE for how long (the captured
# step runs on several streams since round 6: a kernel that runs alone is on the step's critical path, one that always has
# company is hidden)
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]


pts = []
for i, (s, e, n) in enumerate(ev):
    pts.append((s, 1, i))
    pts.append((e, -1, i))
pts.sort(key=lambda p: (p[0], p[1]))
live, last = set(), pts[0][0]
conc = defaultdict(int)
alone = defaultdict(int)
for t, d, i in pts:
    if t > last:
        conc[min(len(live), 4)] += t - last
        if len(live) == 1:
            alone[short(ev[next(iter(live))][2])] += t - last
        last = t
    if d > 0:
        live.add(i)
    else:
        live.discard(i)
print("kernels running at once (ms):", {k: round(v / 1e6, 2) for k, v in sorted(conc.items())})
print("time running ALONE, by kernel:")
tot = defaultdict(int)
for s_, e_, n_ in ev:
    tot[short(n_)] += e_ - s_
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:40]:
    print("%8.1f us alone of %8.1f  %s" % (v / 1e3, tot[n] / 1e3, n))
