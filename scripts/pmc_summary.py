#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name (mean per dispatch) and derive
HBM traffic per launch with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE is in KB and
reports 1/2 of the bytes of wide coalesced reads -> x2; WRITE_SIZE in KB, uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "*", "*counter_collection.csv")):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("void ", "").replace("mvconv::", "")
            k = k.split("(")[0]
            a = agg[k][r.get("Counter_Name")]
            a[0] += float(r.get("Counter_Value", 0) or 0)
            a[1] += 1
out = {}
for k in agg:
    m = {c: agg[k][c][0] / max(agg[k][c][1], 1) for c in agg[k]}
    m["dispatches"] = max(v[1] for v in agg[k].values())
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        m["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"] > 0:
        # busy cycles summed over 1024 SIMDs vs GUI-active cycles summed over 8 XCDs
        m["mfma_busy_frac"] = (m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (m["GRBM_GUI_ACTIVE"] / 8.0)
    out[k] = m
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
if len(sys.argv) > 3:
    # compact per-kernel HBM traffic for bench.py's roofline.traffic (our kernels only)
    mine = {k: round(m["hbm_bytes_per_launch"]) for k, m in out.items()
            if "hbm_bytes_per_launch" in m and k.split("<")[0] in (
                "conv_lds_kernel", "conv_mfma_kernel", "conv_pers_kernel", "conv_pers8_kernel", "conv_tpers_kernel", "conv1x1_pers_kernel", "conv_wino_kernel", "conv_wino_ring_kernel",
                "conv_small_kernel", "conv_narrow_kernel", "deconv_small_kernel", "deconv_select_kernel", "deconv_select_mfma_kernel",
                "warp_agg_fwd_kernel", "warp_agg_fwd_lanes_kernel", "warp_agg_fwd_wave_kernel",
                "warp_agg_fwd_pix_kernel", "fpn_tail_gather_lds_kernel", "fpn_tail_fused_kernel", "fpn_lateral_up_kernel")}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash
    with open(sys.argv[3], "w") as f:
        json.dump({"workload": [512, 640, 5], "kernel_source_hash": kernel_source_hash(), "source": "scripts/gpu_pmc.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / "
                   "WRITE_SIZE (separate passes) of bench.py --no-graph; (2*FETCH_SIZE + WRITE_SIZE) KB per launch",
                   "kernels": mine}, f, indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k].get("GRBM_GUI_ACTIVE", 0) * out[k]["dispatches"])[:40]:
    m = out[k]
    print("%-44s n=%4d  hbm/launch %8.2f MB  mfma_busy %5.1f%%  valu/mfma %6.1f  lds_conflict %5.1f%%" % (
        k[:44], m["dispatches"], m.get("hbm_bytes_per_launch", 0) / 1e6, 100 * m.get("mfma_busy_frac", 0),
        m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_INSTS_MFMA", 0), 1),
        100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 0), 1)))
