#!/usr/bin/env python3
"""Persistent LDS-DMA convolution kernel (variant 5, conv_pers.hip): value check against the direct kernel (same K
order -> expected bit-identical) on ragged and production shapes, and timing against the plan's current choice.
GPU only.  Usage: conv_pers_check.py [--time-only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from conv_microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
# (name, cin, cout, kernel, stride, nt)
FAMILIES = [
    ("16->16 3x3", 16, 16, (1, 3, 3), (1, 1, 1), 1),
    ("32->32 3x3", 32, 32, (1, 3, 3), (1, 1, 1), 2),
    ("16->32 5x5 s2", 16, 32, (1, 5, 5), (1, 2, 2), 2),
    ("16->16 3x3x3", 16, 16, (3, 3, 3), (1, 1, 1), 1),
    ("16->32 3x3 s2", 16, 32, (1, 3, 3), (1, 2, 2), 2),
    ("8->16 5x5 s2", 8, 16, (1, 5, 5), (1, 2, 2), 1),
    ("8->16 3x3 s2", 8, 16, (1, 3, 3), (1, 2, 2), 1),
]
EXTRA = [
    ("64->64 3x3", 64, 64, (1, 3, 3), (1, 1, 1), 1),
    ("64->32 3x3", 64, 32, (1, 3, 3), (1, 1, 1), 1),
    ("32->64 5x5 s2", 32, 64, (1, 5, 5), (1, 2, 2), 4),
    ("32->32 3x3x3", 32, 32, (3, 3, 3), (1, 1, 1), 2),
    ("32->64 3x3 s2", 32, 64, (1, 3, 3), (1, 2, 2), 4),
    ("64->64 3x3x3", 64, 64, (3, 3, 3), (1, 1, 1), 4),
]
CHECK_SHAPES = {1: [(2, 1, 70, 100), (1, 1, 4, 33), (3, 1, 64, 64)], 3: [(2, 4, 38, 70), (1, 8, 16, 40), (1, 3, 5, 31)]}
PROD = {  # production shapes of the 512x640x5 forward (input B, D, H, W) per family
    "16->16 3x3": [(5, 1, 256, 320)],
    "32->32 3x3": [(5, 1, 128, 160)],
    "16->32 5x5 s2": [(5, 1, 256, 320)],
    "16->16 3x3x3": [(1, 4, 256, 320), (1, 4, 128, 160), (1, 8, 64, 80), (1, 8, 32, 40)],
    "16->32 3x3 s2": [(1, 4, 256, 320), (1, 4, 128, 160), (1, 8, 64, 80)],
    "8->16 5x5 s2": [(5, 1, 512, 640)],
    "8->16 3x3 s2": [(1, 4, 512, 640), (1, 4, 256, 320), (1, 8, 128, 160), (1, 8, 64, 80)],
    "64->64 3x3": [(5, 1, 64, 80)],
    "64->32 3x3": [(5, 1, 128, 160)],
    "32->64 5x5 s2": [(5, 1, 128, 160)],
    "32->32 3x3x3": [(1, 4, 128, 160), (1, 4, 64, 80), (1, 8, 32, 40)],
    "32->64 3x3 s2": [(1, 4, 128, 160), (1, 4, 64, 80)],
    "64->64 3x3x3": [(1, 4, 64, 80), (1, 4, 32, 40)],
}


def make_layer(cin, cout, kernel, stride, relu=True):
    g = torch.Generator(device="cpu").manual_seed(cin * 1000 + cout + kernel[0])
    w = (torch.randn(cout, cin, *kernel, generator=g) * 0.1).to(dev)
    pad = tuple(k // 2 for k in kernel)
    layer = cp.ConvLayer(w, False, stride, pad, relu=relu)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    return layer


def supported(layer, x, nt):
    try:
        layer(x, tiles=(2, nt, 5))
        torch.cuda.synchronize()
        return True
    except RuntimeError:
        return False


def check():
    bad = 0
    for name, cin, cout, kernel, stride, nt in FAMILIES + EXTRA:
        layer = make_layer(cin, cout, kernel, stride)
        for shape in CHECK_SHAPES[kernel[0]]:
            x = torch.randn(*shape, cin, device=dev)
            if not supported(layer, x, nt):
                print("%-16s not built" % name)
                break
            want = layer(x, tiles=(1, 1, 0))
            skip = torch.randn_like(want)
            for wpc in (0, 1, 2, 4, 18):
                got = layer(x, tiles=(2, nt, 5 | (wpc << 8)))
                gs = layer(x, skip=skip, skip_mode=1, tiles=(2, nt, 5 | (wpc << 8)))
                ws = layer(x, skip=skip, skip_mode=1, tiles=(1, 1, 0))
                torch.cuda.synchronize()
                ok = torch.equal(got, want) and torch.equal(gs, ws)
                err = (got - want).abs().max().item()
                if not ok:
                    bad += 1
                print("%-16s in %-18s wpc %d: %s (max |d| %.3g, with skip %.3g)" % (
                    name, "x".join(map(str, shape)), wpc, "bit-identical" if ok else "DIFFERENT", err,
                    (gs - ws).abs().max().item()), flush=True)
    print("value check: %s" % ("all bit-identical to the direct kernel" if bad == 0 else "%d MISMATCHES" % bad))
    return bad


def times():
    for name, cin, cout, kernel, stride, nt in FAMILIES + EXTRA:
        layer = make_layer(cin, cout, kernel, stride)
        for shape in PROD[name]:
            x = torch.randn(*shape, cin, device=dev)
            fl = layer.flops(*shape)
            _, mt0, nt0, _, var0 = layer._geom(*shape, 0)
            base = min(timeit(lambda: layer(x), n=10) for _ in range(2))
            row = "%-16s %-18s plan v%d(%d,%d) %6.1f us %5.1f TF/s |" % (name, "x".join(map(str, shape)), var0, mt0, nt0, base,
                                                                       fl / base / 1e6)
            if supported(layer, x, nt):
                for wpc in (1, 2, 3, 33, 34):               # 32 + n: waves 4-7 issue the DMA (stride-2 families)
                    try:
                        us = min(timeit(lambda: layer(x, tiles=(2, nt, 5 | (wpc << 8))), n=10) for _ in range(2))
                    except RuntimeError:
                        continue
                    ok = torch.equal(layer(x, tiles=(2, nt, 5 | (wpc << 8))), layer(x, tiles=(1, 1, 0)))
                    row += " %s%d %5.1f (%5.1f)%s" % ("L" if wpc > 31 else "w", wpc & 15, us, fl / us / 1e6, "" if ok else " DIFFERENT")
                for wpc in (1, 2):
                    try:
                        us = min(timeit(lambda: layer(x, tiles=(4, nt, 5 | (wpc << 8))), n=10) for _ in range(2))
                        ok = torch.equal(layer(x, tiles=(4, nt, 5 | (wpc << 8))), layer(x, tiles=(1, 1, 0)))
                        row += " mt4w%d %5.1f (%5.1f)%s" % (wpc, us, fl / us / 1e6, "" if ok else " DIFFERENT")
                    except RuntimeError:
                        break
            else:
                row += " (not built)"
            print(row, flush=True)


ONE = [  # (name, cin, cout, production input shape, skip)
    ("64->144 1x1", 64, 144, (5, 1, 128, 160), False),
    ("64->72 1x1", 64, 72, (5, 1, 128, 160), False),
    ("64->64 1x1", 64, 64, (5, 1, 64, 80), False),
    ("32->64 1x1", 32, 64, (5, 1, 128, 160), False),
    ("64->64 1x1 +skip", 64, 64, (5, 1, 128, 160), True),
]


def one_by_one():
    """variant 6 (persistent 1x1, weights in LDS) against the direct kernel: values on a ragged shape, time on the
    production shape."""
    bad = 0
    for name, cin, cout, shape, with_skip in ONE:
        layer = make_layer(cin, cout, (1, 1, 1), (1, 1, 1), relu=not with_skip)
        xr = torch.randn(2, 1, 37, 53, cin, device=dev)
        want = layer(xr, tiles=(1, 1, 0))
        sk = torch.randn_like(want) if with_skip else None
        if with_skip:
            want = layer(xr, skip=sk, skip_mode=1, tiles=(1, 1, 0))
        for mt in (1, 2):
            got = layer(xr, skip=sk, skip_mode=1 if with_skip else 0, tiles=(mt, 1, 6))
            torch.cuda.synchronize()
            ok = torch.equal(got, want)
            bad += 0 if ok else 1
            print("%-18s ragged 2x1x37x53 mt%d: %s (max |d| %.3g)" % (name, mt, "bit-identical" if ok else "DIFFERENT",
                                                                      (got - want).abs().max().item()))
        x = torch.randn(*shape, cin, device=dev)
        skp = torch.randn(*shape, cout, device=dev) if with_skip else None
        sm = 1 if with_skip else 0
        fl = layer.flops(*shape)
        _, mt0, nt0, _, var0 = layer._geom(*shape, sm)
        base = min(timeit(lambda: layer(x, skip=skp, skip_mode=sm), n=10) for _ in range(2))
        row = "%-18s %-16s plan v%d(%d,%d) %6.1f us %5.1f TF/s |" % (name, "x".join(map(str, shape)), var0, mt0, nt0, base, fl / base / 1e6)
        for mt in (1, 2):
            for wpc in (1, 2, 3):
                us = min(timeit(lambda: layer(x, skip=skp, skip_mode=sm, tiles=(mt, 1, 6 | (wpc << 8))), n=10) for _ in range(2))
                row += " mt%dw%d %5.1f" % (mt, wpc, us)
        print(row, flush=True)
    print("1x1 value check: %s" % ("all bit-identical to the direct kernel" if bad == 0 else "%d MISMATCHES" % bad))
    return bad


if __name__ == "__main__":
    if "--one" in sys.argv:
        sys.exit(1 if one_by_one() else 0)
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check()
    times()
    sys.exit(1 if rc else 0)
