#!/usr/bin/env python3
"""Phase timeline of conv_lds_kernel from s_memtime stamps (probe build: make -C mvster_amd/csrc timeline).

    MVSTER_LIB=mvster_amd/csrc/libmvster_hip_tl.so python scripts/conv_timeline.py

Per wavefront: t0 kernel entry, t1 global loads of the first chunk issued (+ address setup), t2 loads arrived and
written to LDS, t3 past the barrier, t4 MFMA loop done, t5 epilogue stores issued.  Prints the distribution of every
phase, the lifetime of a workgroup, how many workgroups a CU holds over time and the MFMA-phase share.  GPU only."""
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
if not hasattr(lib, "mvster_debug_timeline"):
    raise SystemExit("load the probe build: MVSTER_LIB=mvster_amd/csrc/libmvster_hip_tl.so")
lib.mvster_debug_timeline.argtypes = [ctypes.c_void_p]
lib.mvster_debug_timeline.restype = ctypes.c_int

CASES = [  # (cin, cout, kernel, stride, input [B,D,H,W], tiles)
    ("16->16 3x3 5x256x320", 16, 16, (1, 3, 3), (1, 1, 1), (5, 1, 256, 320), (2, 1, 1)),
    ("32->32 3x3 5x128x160", 32, 32, (1, 3, 3), (1, 1, 1), (5, 1, 128, 160), (2, 1, 1)),
    ("64->64 3x3 5x64x80", 64, 64, (1, 3, 3), (1, 1, 1), (5, 1, 64, 80), (2, 4, 1)),
    ("16->16 3x3x3 1x4x256x320", 16, 16, (3, 3, 3), (1, 1, 1), (1, 4, 256, 320), (2, 1, 1)),
    ("16->32 5x5 s2 5x256x320", 16, 32, (1, 5, 5), (1, 2, 2), (5, 1, 256, 320), (2, 1, 1)),
]


def run(name, cin, cout, kernel, stride, shape, tiles):
    w = torch.randn(cout, cin, *kernel, device=dev) * 0.1
    pad = tuple(k // 2 for k in kernel)
    layer = cp.ConvLayer(w, False, stride, pad, relu=True)
    x = torch.randn(*shape, cin, device=dev)
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
    mt = tiles[0]
    _, Do, Ho, Wo = layer.out_shape(*shape)
    blocks = -(-Wo // 32) * -(-Ho // (2 * mt)) * Do * shape[0] * (layer.ntile_total // tiles[1])
    buf = torch.zeros(blocks * 4 * 8, dtype=torch.int64, device=dev)
    assert lib.mvster_debug_timeline(buf.data_ptr()) == 0
    layer(x, tiles=tiles)
    torch.cuda.synchronize()
    assert lib.mvster_debug_timeline(None) == 0
    t = buf.cpu().numpy().reshape(blocks, 4, 8).astype(np.int64)
    ts = t[:, :, :6]
    if (ts == 0).any():
        print(name, ": %d waves did not stamp (grid mismatch?)" % int((ts[:, :, 0] == 0).sum()))
        ts = ts[(ts != 0).all(axis=(1, 2))]
    span = ts[:, :, 5].max() - ts[:, :, 0].min()
    print("== %s  tiles %s  %d workgroups: kernel %.1f us (10-launch average), stamped span %d ticks -> %.1f ticks/us"
          % (name, tiles, blocks, us, span, span / us))
    ph = np.diff(ts, axis=2)      # [blocks, 4, 5]
    names = ["prologue + issue loads", "wait loads + LDS store", "barrier", "MFMA loop (all chunks)", "epilogue"]
    life = (ts[:, :, 5] - ts[:, :, 0])
    for i, n in enumerate(names):
        v = ph[:, :, i].ravel()
        print("   %-26s mean %7.0f  p10 %7.0f  median %7.0f  p90 %7.0f  (%.0f%% of a wave's life)"
              % (n, v.mean(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), 100 * v.mean() / life.mean()))
    print("   %-26s mean %7.0f  p10 %7.0f  median %7.0f  p90 %7.0f" % ("wave lifetime", life.mean(), np.percentile(life, 10),
                                                                        np.median(life), np.percentile(life, 90)))
    # residency: workgroups alive per CU over time (HW_ID: cu_id bits 11:8, sh 12, se 15:13; XCC_ID low bits)
    hw, xcc = t[:, 0, 6], t[:, 0, 7] & 0xf
    cu = (xcc << 8) | ((hw >> 8) & 0xff)
    start, end = ts[:, :, 0].min(axis=1), ts[:, :, 5].max(axis=1)
    ncu = len(np.unique(cu))
    occ = []
    for c in np.unique(cu)[:64]:
        m = cu == c
        ev = sorted([(s, 1) for s in start[m]] + [(e, -1) for e in end[m]])
        cur, last, area = 0, ev[0][0], 0
        for tt, d in ev:
            area += cur * (tt - last)
            last = tt
            cur += d
        occ.append(area / max(1, ev[-1][0] - ev[0][0]))
    print("   %d distinct CUs seen; mean resident workgroups per CU while it is busy: %.2f; workgroups per CU %.1f"
          % (ncu, float(np.mean(occ)), blocks / ncu))
    # first-wave start spread (dispatch ramp) and tail
    s0 = np.sort(start - start.min())
    print("   dispatch: 50%% of workgroups started by tick %d, 90%% by %d, last at %d of %d" % (
        s0[len(s0) // 2], s0[int(len(s0) * 0.9)], s0[-1], span))


for c in CASES:
    run(*c)
