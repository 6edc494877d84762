#!/usr/bin/env python3
"""Ping-pong persistent convolution kernel (variant 7, conv_pers.hip): value check against the direct kernel (same K
order -> expected bit-identical) on ragged and production shapes, and timing against the plan's choice and variant 5.
GPU only.  Usage: conv_pp_check.py [--time-only]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from conv_microbench import timeit  # noqa: E402
from conv_pers_check import CHECK_SHAPES, FAMILIES, PROD, dev, make_layer  # noqa: E402

EXTRA_SHAPES = [(5, 1, 256, 320), (1, 1, 8, 32), (1, 1, 5, 200), (7, 1, 12, 64)]


def supported(layer, x, nt):
    try:
        layer(x, tiles=(2, nt, 7))
        torch.cuda.synchronize()
        return True
    except RuntimeError:
        return False


def check():
    bad = 0
    for name, cin, cout, kernel, stride, nt in FAMILIES:
        layer = make_layer(cin, cout, kernel, stride)
        for shape in CHECK_SHAPES[kernel[0]] + (EXTRA_SHAPES if kernel[0] == 1 else []):
            x = torch.randn(*shape, cin, device=dev)
            if not supported(layer, x, nt):
                print("%-16s not built" % name)
                break
            want = layer(x, tiles=(1, 1, 0))
            skip = torch.randn_like(want)
            got = layer(x, tiles=(2, nt, 7))
            gs = layer(x, skip=skip, skip_mode=1, tiles=(2, nt, 7))
            ws = layer(x, skip=skip, skip_mode=1, tiles=(1, 1, 0))
            torch.cuda.synchronize()
            ok = torch.equal(got, want) and torch.equal(gs, ws)
            bad += 0 if ok else 1
            print("%-16s in %-18s: %s (max |d| %.3g, with skip %.3g)" % (
                name, "x".join(map(str, shape)), "bit-identical" if ok else "DIFFERENT", (got - want).abs().max().item(),
                (gs - ws).abs().max().item()), flush=True)
    print("value check: %s" % ("all bit-identical to the direct kernel" if bad == 0 else "%d MISMATCHES" % bad))
    return bad


def times():
    for name, cin, cout, kernel, stride, nt in FAMILIES:
        layer = make_layer(cin, cout, kernel, stride)
        for shape in PROD[name]:
            x = torch.randn(*shape, cin, device=dev)
            if not supported(layer, x, nt):
                continue
            fl = layer.flops(*shape)
            _, mt0, nt0, _, var0 = layer._geom(*shape, 0)
            base = min(timeit(lambda: layer(x), n=10) for _ in range(2))
            row = "%-16s %-18s plan v%d(%d,%d) %6.1f us %5.1f TF/s |" % (name, "x".join(map(str, shape)), var0, mt0, nt0, base,
                                                                       fl / base / 1e6)
            for label, var in (("pers w2", 5 | (2 << 8)), ("pp", 7)):
                us = min(timeit(lambda: layer(x, tiles=(2, nt, var)), n=10) for _ in range(2))
                row += " %s %5.1f (%5.1f TF/s)" % (label, us, fl / us / 1e6)
            print(row, flush=True)


if __name__ == "__main__":
    rc = 0
    if "--time-only" not in sys.argv:
        rc = check()
    times()
    sys.exit(1 if rc else 0)
