#!/usr/bin/env python3
"""Half-phase timeline of conv_pp_kernel (probe build):
    MVSTER_LIB=mvster_amd/csrc/libmvster_hip_tl.so python scripts/conv_pp_timeline.py
Per half-phase: duration of the MFMA half's work, of the preparing half's work (issue, then wait for the patch), and
the half-phase period.  GPU only."""
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
]
NREC = 1 << 12


def stats(name, v):
    v = np.asarray(v)
    print("   %-34s mean %7.0f  p10 %7.0f  median %7.0f  p90 %7.0f" % (name, v.mean(), np.percentile(v, 10), np.median(v),
                                                                     np.percentile(v, 90)))


for name, cin, cout, kernel, stride, shape, nt in CASES:
    w = torch.randn(cout, cin, *kernel, device=dev) * 0.1
    layer = cp.ConvLayer(w, False, stride, tuple(k // 2 for k in kernel), relu=True)
    x = torch.randn(*shape, cin, device=dev)
    tiles = (2, nt, 7)
    for _ in range(3):
        layer(x, tiles=tiles)
    torch.cuda.synchronize()
    buf = torch.zeros(NREC * 8 * 16 * 4, dtype=torch.int64, device=dev)
    assert lib.mvster_debug_pers_timeline(buf.data_ptr()) == 0
    layer(x, tiles=tiles)
    torch.cuda.synchronize()
    assert lib.mvster_debug_pers_timeline(None) == 0
    t = buf.cpu().numpy().reshape(NREC, 8, 16, 4)
    t = t[t[:, 0, 0, 0] != 0]
    print("== %s: %d workgroups" % (name, len(t)))
    mf, prep_issue, prep_wait, period = [], [], [], []
    for g in range(len(t)):
        for hp in range(1, 15):
            if t[g, 0, hp + 1, 0] == 0 or t[g, 0, hp, 0] == 0:
                break
            period.append(t[g, :, hp + 1, 0].max() - t[g, :, hp, 0].max())
            for w8 in range(8):
                r = t[g, w8, hp]
                if ((hp & 1) == (w8 >> 2)):
                    if r[2] == 1:
                        mf.append(r[1] - r[0])
                elif r[2] > 2:
                    prep_issue.append(r[2] - r[0])
                    prep_wait.append(r[1] - r[2])
    stats("MFMA half: top -> done", mf)
    stats("prepare half: top -> all issued", prep_issue)
    stats("prepare half: wait for the patch", prep_wait)
    stats("half-phase period", period)
    g = 0
    print("   workgroup 0, wave 0 / wave 4, top-of-half-phase deltas:", np.diff(t[g, 0, :12, 0]), np.diff(t[g, 4, :12, 0]))
