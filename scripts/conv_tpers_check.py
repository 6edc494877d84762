#!/usr/bin/env python3
"""Persistent transposed-convolution kernel (conv_tpers_kernel, variant 5 on a transposed 1x3x3 stride-(1,2,2) layer): value
check against the direct kernel (same K order per parity class -> bit-identical) and timing.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from conv_microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
FAMILIES = [("T64->32", 64, 32), ("T32->16", 32, 16), ("T64->64", 64, 64)]
CHECK = [(2, 3, 9, 21), (1, 1, 4, 33), (1, 4, 16, 40), (3, 1, 5, 64)]
PROD = {"T64->32": [(1, 4, 64, 80), (1, 4, 32, 40), (1, 8, 16, 20), (1, 8, 8, 10), (2, 4, 64, 80)],
        "T32->16": [(1, 4, 128, 160), (1, 4, 64, 80), (1, 8, 32, 40), (1, 8, 16, 20), (2, 4, 128, 160)], "T64->64": [(1, 4, 64, 80)]}


def make_layer(cin, cout):
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout)
    w = (torch.randn(cin, cout, 1, 3, 3, generator=g) * 0.1).to(dev)
    layer = cp.ConvLayer(w, True, (1, 2, 2), (0, 1, 1), relu=True)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    return layer


bad = 0
for name, cin, cout in FAMILIES:
    layer = make_layer(cin, cout)
    for shape in CHECK:
        x = torch.randn(*shape, cin, device=dev)
        want = layer(x, tiles=(1, 1, 0))
        skip = torch.randn_like(want)
        ws = layer(x, skip=skip, skip_mode=1, tiles=(1, 1, 0))
        for wpc in (0, 1, 2):
            got = layer(x, tiles=(2, 1, 5 | (wpc << 8)))
            gs = layer(x, skip=skip, skip_mode=1, tiles=(2, 1, 5 | (wpc << 8)))
            torch.cuda.synchronize()
            ok = torch.equal(got, want) and torch.equal(gs, ws)
            bad += 0 if ok else 1
            print("%-8s in %-12s wpc %d: %s (max |d| %.3g, with skip %.3g)" % (name, "x".join(map(str, shape)), wpc,
                  "bit-identical" if ok else "DIFFERENT", (got - want).abs().max().item(), (gs - ws).abs().max().item()), flush=True)
print("value check: %s" % ("all bit-identical to the direct kernel" if bad == 0 else "%d MISMATCHES" % bad))
for name, cin, cout in FAMILIES:
    layer = make_layer(cin, cout)
    for shape in PROD[name]:
        x = torch.randn(*shape, cin, device=dev)
        B, D, H, W = shape
        skip = torch.randn(B, D, 2 * H, 2 * W, cout, device=dev)
        fl = layer.flops(*shape)
        _, mt0, nt0, _, var0 = layer._geom(*shape, 1)
        base = min(timeit(lambda: layer(x, skip=skip, skip_mode=1), n=10) for _ in range(2))
        row = "%-8s %-14s plan v%d(%d,%d) %6.1f us %5.1f TF/s |" % (name, "x".join(map(str, shape)), var0, mt0, nt0, base, fl / base / 1e6)
        for wpc in (1, 2):
            us = min(timeit(lambda: layer(x, skip=skip, skip_mode=1, tiles=(2, 1, 5 | (wpc << 8))), n=10) for _ in range(2))
            row += " w%d %5.1f (%5.1f)" % (wpc, us, fl / us / 1e6)
        print(row, flush=True)
sys.exit(1 if bad else 0)
