#!/usr/bin/env python3
"""Group a rocprofv3 kernel_stats.csv of training steps into kernel families: ms and launches per step.
usage: train_categories.py <kernel_stats.csv> <steps in the profile> [<kernel_stats.csv of a SHORTER run> <its steps>]
With the second pair the shorter run is subtracted kernel by kernel and the difference divided by the difference of the step
counts: start-up work (weight upload, first-call layer builds, allocator warm-up: ~400 fills and copies) then does not
count as per-step work."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
if len(sys.argv) > 4:
    short = {r["Name"]: r for r in csv.DictReader(open(sys.argv[3]))}
    steps -= float(sys.argv[4])
    kept = []
    for r in rows:
        b = short.get(r["Name"])
        t = float(r["TotalDurationNs"]) - (float(b["TotalDurationNs"]) if b else 0.0)
        c = int(r["Calls"]) - (int(b["Calls"]) if b else 0)
        if c > 0 and t > 0:
            kept.append(dict(r, TotalDurationNs=t, Calls=c))
    rows = kept
    print("  (steady state: %s minus %s, %d steps)" % (sys.argv[1].split("/")[-1], sys.argv[3].split("/")[-1], int(steps)))
cat, cnt = collections.Counter(), collections.Counter()
for r in rows:
    n = r["Name"]
    t, c = float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps
    if "wgrad" in n: k = "weight gradients (+finish)"
    elif "conv_" in n or "deconv" in n or "conv1x1" in n: k = "conv forward / input gradient"
    elif "bn_" in n[:60]: k = "BatchNorm kernels"
    elif "warp_agg_bwd" in n or "scatter_gather" in n or "absmax" in n: k = "warp backward"
    elif "warp_agg_fwd" in n: k = "warp forward"
    elif "select_depth" in n: k = "stage selection (fwd + bwd)"
    elif "fpn_" in n: k = "FPN gather / adjoint"
    elif "at::native" in n or "rocclr" in n or "Cijk" in n or "rocblas" in n or "at::" in n: k = "torch (aten) kernels"
    elif "pack_weights" in n: k = "weight re-pack"
    elif "sinkhorn" in n: k = "sinkhorn"
    elif "upsample" in n: k = "upsample"
    else: k = "other: " + re.sub(r"\(anonymous namespace\)::", "", n)[:40]
    cat[k] += t
    cnt[k] += c
for k, v in cat.most_common(16):
    print("  %6.2f ms %6.0f launches  %s" % (v, cnt[k], k))
print("  total %.2f ms, %d launches per step" % (sum(cat.values()), sum(cnt.values())))
aten = [(float(r["TotalDurationNs"]) / steps / 1e3, int(r["Calls"]) / steps, r["Name"]) for r in rows
        if "at::" in r["Name"] or "rocclr" in r["Name"]]
aten.sort(reverse=True)
for t, c, n in aten[:14]:
    print("     %7.1f us %5.0f x %s" % (t, c, n[:150]))
