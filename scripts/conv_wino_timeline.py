#!/usr/bin/env python3
"""Step timeline of conv_wino_ring_kernel (probe build):
    MVSTER_LIB=mvster_amd/csrc/libmvster_hip_tl.so python scripts/conv_wino_timeline.py
Per step and wave: MFMA phase, transform (+ epilogue), end-of-step wait + barrier, in s_memtime ticks.  GPU only."""
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
if not hasattr(lib, "mvster_debug_wino_timeline"):
    raise SystemExit("load the probe build: MVSTER_LIB=mvster_amd/csrc/libmvster_hip_tl.so")
lib.mvster_debug_wino_timeline.argtypes = [ctypes.c_void_p]
lib.mvster_debug_wino_timeline.restype = ctypes.c_int
CASES = [("64->64 3x3x3 nt1 1x8x8x10", 64, 64, 3, 1, (1, 8, 8, 10)), ("64->64 3x3x3 nt2 1x4x64x80", 64, 64, 3, 2, (1, 4, 64, 80)),
         ("64->64 3x3x3 nt1 1x4x64x80", 64, 64, 3, 1, (1, 4, 64, 80)), ("16->16 3x3x3 nt1 1x4x256x320", 16, 16, 3, 1, (1, 4, 256, 320))]
NREC = 1 << 11


def stats(name, v):
    v = np.asarray(v)
    print("   %-44s mean %7.0f  p10 %7.0f  median %7.0f  p90 %7.0f" % (name, v.mean(), np.percentile(v, 10), np.median(v),
                                                                     np.percentile(v, 90)))


for name, cin, cout, kd, nt, shape in CASES:
    w = torch.randn(cout, cin, kd, 3, 3, device=dev) * 0.1
    layer = cp.ConvLayer(w, False, (1, 1, 1), (kd // 2, 1, 1), relu=True)
    x = torch.randn(*shape, cin, device=dev)
    tiles = (2, nt, 9)
    for _ in range(3):
        layer(x, tiles=tiles)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf = torch.zeros(NREC * 8 * 32 * 4, dtype=torch.int64, device=dev)
    assert lib.mvster_debug_wino_timeline(buf.data_ptr()) == 0
    e0.record()
    layer(x, tiles=tiles)
    e1.record()
    torch.cuda.synchronize()
    assert lib.mvster_debug_wino_timeline(None) == 0
    t = buf.cpu().numpy().reshape(NREC, 8, 32, 4)
    t = t[t[:, 0, 0, 0] != 0]
    print("== %s: %d workgroups, probed launch %.1f us" % (name, len(t), e0.elapsed_time(e1) * 1e3))
    comp = [0, 1, 2, 3] if nt == 1 else list(range(8))
    mf, tr, wb, per, ld_issue, ld_wait, ld_bar = [], [], [], [], [], [], []
    for g in range(len(t)):
        for st in range(1, 31):
            if t[g, 0, st + 1, 0] == 0:
                break
            for w8 in comp:
                r = t[g, w8, st]
                mf.append(r[1] - r[0])
                tr.append(r[2] - r[1])
                wb.append(r[3] - r[2])
                per.append(t[g, w8, st + 1, 0] - r[0])
            if nt == 1:
                for w8 in range(4, 8):
                    r = t[g, w8, st]
                    ld_issue.append(r[1] - r[0])
                    ld_wait.append(r[2] - r[1])
                    ld_bar.append(r[3] - r[2])
    stats("compute: top -> MFMAs issued", mf)
    stats("compute: nops, (epilogue,) transform", tr)
    stats("compute: end-of-step wait + barrier", wb)
    stats("compute: step period", per)
    if nt == 1:
        stats("loader: issue weights + slice requests", ld_issue)
        stats("loader: wait for landing", ld_wait)
        stats("loader: barrier", ld_bar)
    print("   workgroup 0 wave 0 step tops:", np.diff(t[0, 0, :14, 0]))
