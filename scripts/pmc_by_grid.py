#!/usr/bin/env python3
"""Per (kernel, grid size): mean duration from the kernel trace and mean PMC counters of a set of rocprofv3 passes."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").replace("mvconv::", "").split("(")[0]


dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "*kernel_trace.csv")):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            dur[(short(r["Kernel_Name"]), r.get("Grid_Size") or r.get("Grid_Size_X"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            cnt[(short(r["Kernel_Name"]), r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"] or 0))
for key in sorted(cnt, key=lambda k: k[0]):
    if not any(key[0].startswith(p) for p in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("conv_narrow", "conv_small", "deconv_select"))):
        continue
    m = {c: sum(v[1:]) / max(len(v) - 1, 1) for c, v in cnt[key].items()}          # (first launch of a case dropped)
    d = sorted(dur.get(key, [0]))
    us = d[len(d) // 2] / 1e3
    g = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    wc = max(m.get("SQ_WAVE_CYCLES", 0), 1)
    print("%-46s grid %-8s %6.1f us | clk %.2f GHz  mfma_busy %4.1f%%  busy %4.1f%% | of wave cycles: wait_any %4.1f%% wait_inst %4.1f%% active %4.1f%% | "
          "valu %d mfma %d waves %d | hbm %.1f MB lds_conf %.1f%%" % (
              key[0][:46], key[1], us, g / max(us, 1e-9) / 1e3, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(g, 1),
              100 * m.get("SQ_BUSY_CYCLES", 0) / 32 / max(g, 1),
              100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
              m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_MFMA", 0), m.get("SQ_WAVES", 0),
              (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024 / 1e6,
              100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1)))
