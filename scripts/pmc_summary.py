#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "?")
            c = r.get("Counter_Name")
            v = float(r.get("Counter_Value", 0) or 0)
            a = agg[k][c]
            a[0] += v
            a[1] += 1
names = sorted({c for k in agg for c in agg[k]})
print("kernel | " + " | ".join(names))
def key(k):
    return -agg[k].get("SQ_BUSY_CYCLES", [0, 1])[0]
for k in sorted(agg, key=key)[:45]:
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(short + " | " + " | ".join("%s=%.4g(n%d)" % (c, agg[k][c][0] / max(agg[k][c][1], 1), agg[k][c][1]) if c in agg[k] else "-" for c in names))
