#!/usr/bin/env python3
"""Winograd F(2x2,3x3) convolution kernel (variant 8, conv_wino.hip): error against an fp64 convolution next to the
direct kernel's own error, on ragged and production shapes, and timing against the plan's choice.
GPU only.  Usage: conv_wino_check.py [--time-only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from conv_microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
# (name, cin, cout, kd, nt, variant)
FAMILIES = [("16->16 3x3", 16, 16, 1, 1, 8), ("16->32 3x3", 16, 32, 1, 2, 8), ("32->32 3x3", 32, 32, 1, 2, 8),
            ("32->16 3x3", 32, 16, 1, 1, 8),
            ("16->16 3x3 ring", 16, 16, 1, 1, 9), ("32->32 3x3 ring", 32, 32, 1, 2, 9), ("64->64 3x3 ring", 64, 64, 1, 2, 9),
            ("64->64 3x3 ring1", 64, 64, 1, 1, 9), ("64->32 3x3 ring", 64, 32, 1, 2, 9), ("64->32 3x3 ring1", 64, 32, 1, 1, 9),
            ("16->16 3x3x3", 16, 16, 3, 1, 9), ("32->32 3x3x3", 32, 32, 3, 2, 9), ("32->32 3x3x3 nt1", 32, 32, 3, 1, 9),
            ("64->64 3x3x3", 64, 64, 3, 2, 9), ("64->64 3x3x3 nt1", 64, 64, 3, 1, 9),
            # mode 2 of the ring kernel (variant word 9 | 1 << 8): both N tiles in the compute waves, waves 4-7 load
            ("32->32 3x3 m2", 32, 32, 1, 2, 265), ("64->64 3x3 m2", 64, 64, 1, 2, 265), ("64->32 3x3 m2", 64, 32, 1, 2, 265),
            ("16->32 3x3 m2", 16, 32, 1, 2, 265), ("32->32 3x3x3 m2", 32, 32, 3, 2, 265), ("64->64 3x3x3 m2", 64, 64, 3, 2, 265)]
CHECK_SHAPES = [(2, 1, 70, 100), (1, 1, 4, 33), (3, 1, 64, 64), (1, 1, 8, 32), (1, 1, 5, 200), (7, 1, 13, 63), (1, 4, 38, 70),
                (2, 3, 9, 40), (1, 8, 16, 20)]
PROD = {"16->16 3x3": [(5, 1, 256, 320)], "32->32 3x3": [(5, 1, 128, 160)], "16->32 3x3": [(5, 1, 256, 320)],
        "32->16 3x3": [(5, 1, 128, 160)], "16->16 3x3 ring": [(5, 1, 256, 320)], "32->32 3x3 ring": [(5, 1, 128, 160)],
        "64->64 3x3 ring": [(5, 1, 64, 80)], "64->64 3x3 ring1": [(5, 1, 64, 80)], "64->32 3x3 ring": [(5, 1, 128, 160)],
        "64->32 3x3 ring1": [(5, 1, 128, 160)],
        "16->16 3x3x3": [(1, 4, 256, 320), (1, 4, 128, 160), (1, 8, 64, 80), (1, 8, 32, 40)],
        "32->32 3x3x3": [(1, 4, 128, 160), (1, 4, 64, 80), (1, 8, 32, 40), (1, 8, 16, 20)],
        "32->32 3x3x3 nt1": [(1, 4, 128, 160), (1, 4, 64, 80), (1, 8, 32, 40), (1, 8, 16, 20)],
        "64->64 3x3x3": [(1, 4, 64, 80), (1, 4, 32, 40), (1, 8, 16, 20), (1, 8, 8, 10)],
        "64->64 3x3x3 nt1": [(1, 4, 64, 80), (1, 4, 32, 40), (1, 8, 16, 20), (1, 8, 8, 10)],
        "32->32 3x3 m2": [(5, 1, 128, 160)], "64->64 3x3 m2": [(5, 1, 64, 80)], "64->32 3x3 m2": [(5, 1, 128, 160)],
        "16->32 3x3 m2": [(5, 1, 256, 320)], "32->32 3x3x3 m2": [(1, 4, 128, 160), (1, 4, 64, 80)],
        "64->64 3x3x3 m2": [(1, 4, 64, 80), (1, 4, 32, 40)]}


def make_layer(cin, cout, kd=1, relu=True):
    g = torch.Generator(device="cpu").manual_seed(cin * 1000 + cout + kd)
    w = (torch.randn(cout, cin, kd, 3, 3, generator=g) * 0.1).to(dev)
    layer = cp.ConvLayer(w, False, (1, 1, 1), (kd // 2, 1, 1), relu=relu)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    return layer, w


def exact(layer, w, x, skip):
    """fp64 reference of the fused layer on channels-last x [B,D,H,W,C]."""
    y = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), padding=(w.shape[2] // 2, 1, 1))
    y = y * layer.scale[:layer.cout].double().view(1, -1, 1, 1, 1) + layer.shift[:layer.cout].double().view(1, -1, 1, 1, 1)
    if layer.relu:
        y = y.clamp_min(0)
    y = y.permute(0, 2, 3, 4, 1)
    return y + skip.double() if skip is not None else y


def check():
    bad = 0
    for name, cin, cout, kd, nt, var in FAMILIES:
        layer, w = make_layer(cin, cout, kd)
        for shape in CHECK_SHAPES:
            x = torch.randn(*shape, cin, device=dev)
            skip = torch.randn(*shape, cout, device=dev)
            for sk in (None, skip):
                sm = 0 if sk is None else 1
                ref = exact(layer, w, x, sk)
                direct = layer(x, skip=sk, skip_mode=sm, tiles=(1, 1, 0))
                for wpc in ((1, 2) if var == 8 else (0,)):
                    got = layer(x, skip=sk, skip_mode=sm, tiles=(2, nt, var | (wpc << 8)))        # (var may carry its own mode bits)
                    torch.cuda.synchronize()
                    scale = ref.abs().max().item()
                    e_w = (got.double() - ref).abs().max().item() / scale
                    e_d = (direct.double() - ref).abs().max().item() / scale
                    ok = e_w < 2e-6 and torch.isfinite(got).all().item()
                    bad += 0 if ok else 1
                    print("%-17s in %-14s skip %d wpc %d: winograd err %.2e, direct err %.2e (of max |y|) %s" % (
                        name, "x".join(map(str, shape)), sm, wpc, e_w, e_d, "ok" if ok else "BAD"), flush=True)
    print("value check: %s" % ("all within 2e-6 of max |y| of the fp64 result" if bad == 0 else "%d FAILURES" % bad))
    return bad


def times():
    for name, cin, cout, kd, nt, var in FAMILIES:
        layer, _ = make_layer(cin, cout, kd)
        for shape in PROD[name]:
            x = torch.randn(*shape, cin, device=dev)
            fl = layer.flops(*shape)
            _, mt0, nt0, _, var0 = layer._geom(*shape, 0)
            base = min(timeit(lambda: layer(x), n=10) for _ in range(2))
            row = "%-17s %-14s plan v%d(%d,%d) %6.1f us %5.1f TF/s |" % (name, "x".join(map(str, shape)), var0, mt0, nt0, base,
                                                                       fl / base / 1e6)
            for wpc in ((1, 2, 3) if var == 8 else (0,)):
                try:
                    us = min(timeit(lambda: layer(x, tiles=(2, nt, var | (wpc << 8))), n=10) for _ in range(2))
                except RuntimeError:
                    continue
                row += " wino w%d %5.1f (%5.1f TF/s alg.)" % (wpc, us, fl / us / 1e6)
            print(row, flush=True)


if __name__ == "__main__":
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check()
    times()
    sys.exit(1 if rc else 0)
