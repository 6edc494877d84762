#!/usr/bin/env python3
"""Narrow full-resolution layers (Cin in {4, 8} -> 8, 3x3): the shift-packed MFMA kernel on the LDS-DMA ring (variant 10,
conv_narrow.hip) against the VALU kernel (variant 3) -- values (against an fp64 convolution) and isolated hipGraph timings on
the shapes of the three inference workloads.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from mvster_amd import _lib  # noqa: E402
from scripts.conv_microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")


def layer_for(cin, relu=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(8, cin, 1, 3, 3, generator=g) * 0.2
    layer = cp.ConvLayer(w.to(dev), False, (1, 1, 1), (0, 1, 1), bias=torch.randn(8, generator=g).to(dev), relu=relu,
                         cin_pad=cin)
    return layer, w


def values():
    worst = 0.0
    for cin in (4, 8):
        for (B, D, H, W) in ((1, 1, 8, 32), (2, 3, 21, 45), (1, 4, 37, 130), (5, 1, 64, 96), (1, 2, 5, 7)):
            for relu in (True, False):
                layer, w = layer_for(cin, relu, seed=cin + H)
                x = torch.randn(B, D, H, W, cin, device=dev)
                skip = torch.randn(B, D, H, W, 8, device=dev)
                for sk in (None, skip):
                    ref = F.conv3d(x.permute(0, 4, 1, 2, 3).double().cpu(), w.double(), padding=(0, 1, 1))
                    ref = ref * layer.scale[:8].double().cpu().view(1, 8, 1, 1, 1) + layer.shift[:8].double().cpu().view(1, 8, 1, 1, 1)
                    if relu:
                        ref = ref.clamp_min(0)
                    ref = ref.permute(0, 2, 3, 4, 1)
                    if sk is not None:
                        ref = ref + sk.double().cpu()
                    sm = cp.SKIP_ADD if sk is not None else cp.SKIP_NONE
                    valu = layer(x, skip=sk, skip_mode=sm, tiles=(0, 0, 3))
                    scale = ref.abs().max().item()
                    for mt in (2, 4):
                        for wpc in (1, 2):
                            got = layer(x, skip=sk, skip_mode=sm, tiles=(mt, wpc, 10))
                            assert _lib.last_kernel().startswith("conv_narrow_kernel<%d, %d" % (cin, mt)), _lib.last_kernel()
                            e = (got.double().cpu() - ref).abs().max().item() / scale
                            ev = (got - valu).abs().max().item() / scale
                            worst = max(worst, e)
                            assert e < 2e-6 and ev < 2e-6, (cin, B, D, H, W, relu, sk is not None, mt, wpc, e, ev)
    print("values: worst error against fp64 / max|y| = %.2e (every case also within 2e-6 of the VALU kernel)" % worst)


def timings():
    print("%-34s %8s | %s" % ("shape", "MB", "us (TB/s): VALU, then MFMA mt,wpc"))
    shapes = [(4, 5, 1, 512, 640, False), (8, 5, 1, 512, 640, False), (8, 5, 1, 512, 640, True), (4, 1, 4, 512, 640, False),
              (4, 1, 4, 256, 320, False), (8, 1, 8, 128, 160, False), (8, 1, 8, 64, 80, False),
              (4, 5, 1, 1152, 1600, False), (8, 5, 1, 1152, 1600, True), (8, 10, 1, 512, 640, True), (8, 1, 8, 144, 200, False)]
    for cin, B, D, H, W, sk in shapes:
        layer, _ = layer_for(cin)
        x = torch.randn(B, D, H, W, cin, device=dev)
        skip = torch.randn(B, D, H, W, 8, device=dev) if sk else None
        sm = cp.SKIP_ADD if sk else cp.SKIP_NONE
        mb = B * D * H * W * (cin + 8 + (8 if sk else 0)) * 4 / 1e6
        res = []
        for name, tiles in (("valu", (0, 0, 3)), ("2,1", (2, 1, 10)), ("2,2", (2, 2, 10)), ("4,1", (4, 1, 10)), ("4,2", (4, 2, 10))):
            us = min(timeit(lambda: layer(x, skip=skip, skip_mode=sm, tiles=tiles), n=10) for _ in range(3))
            res.append("%s:%.1f(%.2f)" % (name, us, mb / us))
        print("C%d-8 %dx%dx%dx%d sk%d %17s %8.1f | %s" % (cin, B, D, H, W, int(sk), "", mb, "  ".join(res)), flush=True)


if __name__ == "__main__":
    values()
    timings()
