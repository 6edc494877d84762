#!/usr/bin/env python3
"""Per-tile phase timeline of conv_pers_kernel (probe build):
    MVSTER_LIB=mvster_amd/csrc/libmvster_hip_tl.so python scripts/conv_pers_timeline.py
For workgroups-per-CU 1..4: where the workgroups were placed (CUs used, workgroups per CU) and the per-tile phases
(issue DMA + addresses / MFMA loop / wait + barrier / epilogue) in shader cycles.  GPU only."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from mvster_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
if not hasattr(lib, "mvster_debug_pers_timeline"):
    raise SystemExit("load the probe build: MVSTER_LIB=mvster_amd/csrc/libmvster_hip_tl.so")
lib.mvster_debug_pers_timeline.argtypes = [ctypes.c_void_p]
lib.mvster_debug_pers_timeline.restype = ctypes.c_int

CASES = [
    ("16->16 3x3 5x256x320", 16, 16, (1, 3, 3), (1, 1, 1), (5, 1, 256, 320), 1),
    ("32->32 3x3 5x128x160", 32, 32, (1, 3, 3), (1, 1, 1), (5, 1, 128, 160), 2),
    ("16->16 3x3x3 1x4x256x320", 16, 16, (3, 3, 3), (1, 1, 1), (1, 4, 256, 320), 1),
]
NREC = 1 << 14        # workgroups the buffer holds


def run(name, cin, cout, kernel, stride, shape, nt):
    w = torch.randn(cout, cin, *kernel, device=dev) * 0.1
    layer = cp.ConvLayer(w, False, stride, tuple(k // 2 for k in kernel), relu=True)
    x = torch.randn(*shape, cin, device=dev)
    for wpc in (1, 2, 3, 4):
        tiles = (2, nt, 5 | (wpc << 8))
        for _ in range(3):
            layer(x, tiles=tiles)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            layer(x, tiles=tiles)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        buf = torch.zeros(NREC * 4 * 16 * 8, dtype=torch.int64, device=dev)
        assert lib.mvster_debug_pers_timeline(buf.data_ptr()) == 0
        layer(x, tiles=tiles)
        torch.cuda.synchronize()
        assert lib.mvster_debug_pers_timeline(None) == 0
        t = buf.cpu().numpy().reshape(NREC, 4, 16, 8)
        used = t[:, 0, 0, 0] != 0
        nwg = int(used.sum())
        t = t[used]
        ntile = (t[:, 0, :, 0] != 0).sum(axis=1)
        hw, xcc = t[:, 0, 0, 5], t[:, 0, 0, 6] & 0xf
        cu = (xcc << 16) | (hw & 0xff00) | ((hw >> 12) & 0xf) << 4      # cu_id 11:8, sh 12, se 15:13 (+ xcc)
        ucu, cnt = np.unique(cu, return_counts=True)
        print("== %s wpc %d: %.1f us (10-launch avg); %d workgroups on %d distinct CUs (per CU: min %d max %d), tiles per "
              "workgroup %d..%d" % (name, wpc, us, nwg, len(ucu), cnt.min(), cnt.max(), ntile.min(), ntile.max()))
        # phases per tile, over waves and tiles with full records (skip the first tile: its patch was loaded up front)
        ph = {k: [] for k in ("addr + DMA issue", "MFMA loop", "wait DMA + barrier", "epilogue", "tile total")}
        for g in range(nwg):
            for it in range(1, min(int(ntile[g]), 16)):
                r = t[g, :, it, :5].astype(np.int64)
                if (r == 0).any():
                    continue
                ph["addr + DMA issue"].append(r[:, 1] - r[:, 0])
                ph["MFMA loop"].append(r[:, 2] - r[:, 1])
                ph["wait DMA + barrier"].append(r[:, 3] - r[:, 2])
                ph["epilogue"].append(r[:, 4] - r[:, 3])
                ph["tile total"].append(r[:, 4] - r[:, 0])
        for k, v in ph.items():
            v = np.concatenate(v)
            print("   %-20s mean %7.0f  p10 %7.0f  median %7.0f  p90 %7.0f" % (k, v.mean(), np.percentile(v, 10), np.median(v),
                                                                             np.percentile(v, 90)))
        life = (t[:, :, :, 4].max(axis=2) - t[:, :, 0, 0]).astype(np.int64)
        print("   workgroup lifetime (first loop top -> last epilogue): mean %d cycles, max %d" % (life.mean(), life.max()))


for c in CASES:
    run(*c)
