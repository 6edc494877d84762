#!/usr/bin/env python3
"""s_memtime timeline of the bf16-split convolution kernel (probe library: make -C mvster_amd/csrc probes; run with
MVSTER_LIB=mvster_amd/csrc/libmvster_hip_probes.so).  Prints, per stage of workgroup 0, the phase durations of one compute
wave and one loading wave in shader cycles."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from mvster_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
lib.mvster_b3_timeline.argtypes = [ctypes.c_void_p]
def word(wpc, prio=1, dbg=0):
    return wpc | (prio << 2) | (dbg << 4)


CASES = [(64, 64, 3, (1, 8, 8, 10), 1, word(1)), (64, 64, 3, (1, 8, 8, 10), 1, word(1, 1, 1)), (64, 64, 3, (1, 8, 8, 10), 1, word(1, 1, 2)),
         (16, 16, 1, (5, 1, 256, 320), 2, word(1)), (16, 16, 1, (5, 1, 256, 320), 1, word(1)), (32, 32, 1, (5, 1, 128, 160), 1, word(1)),
         (64, 64, 1, (5, 1, 64, 80), 1, word(1))]
for cin, cout, kd, (B, D, H, W), tyq, wpc in CASES:
    g = torch.Generator().manual_seed(1)
    w = torch.randn(cout, cin, kd, 3, 3, generator=g) * 0.05
    layer = cp.ConvLayer(w.to(dev), False, (1, 1, 1), (kd // 2, 1, 1), relu=True)
    x = torch.randn(B, D, H, W, cin, generator=g).to(dev)
    word = 11 | (wpc << 8)
    for _ in range(3):
        layer(x, tiles=(tyq, 1, word))
    buf = torch.zeros(4 * 8 * 32 * 4, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    lib.mvster_b3_timeline(buf.data_ptr())
    layer(x, tiles=(tyq, 1, word))
    torch.cuda.synchronize()
    lib.mvster_b3_timeline(None)
    t = buf.cpu().view(4, 8, 32, 4)
    print("== C%d-%d kd%d %dx%dx%dx%d TY%d wpc%d prio%d dbg%d (%s)" % (cin, cout, kd, B, D, H, W, 4 * tyq, wpc & 3, (wpc >> 2) & 3, wpc >> 4, _lib.last_kernel()))
    for wg in (0,):
        c, l = t[wg, 0], t[wg, 4]
        t0 = min(int(v) for v in (c[0, 0], l[0, 0]) if v > 0)
        print(" workgroup %d   stage: compute [barrier-exit, +mfma issued, +epilogue]   loader [top, +dma issued, +split, +barrier]  (cycles from the first stamp)" % wg)
        for k in range(32):
            if c[k, 0] == 0 and l[k, 0] == 0:
                break
            cs = ["%7d" % (int(v) - t0) if v > 0 else "      -" for v in c[k, :3]]
            ls = ["%7d" % (int(v) - t0) if v > 0 else "      -" for v in l[k]]
            mf = int(c[k, 1] - c[k, 0]) if c[k, 1] > 0 else 0
            wr = int(l[k, 2] - l[k, 1]) if l[k, 2] > 0 else 0
            print("   %2d  compute %s  (mfma loop %6d)   loader %s  (split %6d, barrier wait %6d)" % (
                k, " ".join(cs), mf, " ".join(ls), wr, int(l[k, 3] - l[k, 2]) if l[k, 3] > 0 else 0))
