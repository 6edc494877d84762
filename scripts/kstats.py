#!/usr/bin/env python3
"""Print the kernels of a rocprofv3 kernel_stats.csv whose name contains one of the given substrings: kstats.py <csv> [substr...]"""
import csv
import sys

pats = sys.argv[2:] or [""]
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"])):
    if any(p in r["Name"] for p in pats):
        print("%9.1f us avg %5d x  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"].replace("(anonymous namespace)::", "")[:120]))
