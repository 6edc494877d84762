#!/usr/bin/env python3
"""Transposed 1x5x5 stride (1,2,2) layers (the input gradients of the FPN's 5x5 stride-2 convolutions) on the persistent kernel
(conv_tpers_kernel<., ., 5>, tiles = (2, 1, 5)): bit-identical to the direct kernel on ragged and production shapes, with and
without the epilogue's skip, and timing against the direct kernel and the plan's choice.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from mvster_amd import _lib  # noqa: E402
from conv_microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(1, 1, 7, 33), (2, 1, 12, 40), (3, 1, 5, 70), (1, 2, 16, 32)]
PROD = {(16, 8): (10, 1, 256, 320), (32, 16): (10, 1, 128, 160)}


def main():
    bad = 0
    g = torch.Generator().manual_seed(2)
    for cin, cout in ((16, 8), (32, 16), (16, 16), (32, 32)):
        w = (0.1 * torch.randn(cin, cout, 1, 5, 5, generator=g)).to(dev)
        layer = cp.ConvLayer(w, True, (1, 2, 2), (0, 2, 2), relu=False)
        for shape in SHAPES + ([PROD[(cin, cout)]] if (cin, cout) in PROD else []):
            x = torch.randn(*shape, cin, generator=g).to(dev)
            B, D, H, W = shape
            skip = torch.randn(B, D, 2 * H, 2 * W, cout, generator=g).to(dev)
            for sk in (None, skip):
                sm = cp.SKIP_NONE if sk is None else cp.SKIP_ADD
                want = layer(x, skip=sk, skip_mode=sm, tiles=(1, 1, 0))
                got = layer(x, skip=sk, skip_mode=sm, tiles=(2, 1, 5))
                name = _lib.last_kernel()
                torch.cuda.synchronize()
                same = torch.equal(want, got) and name.startswith("conv_tpers_kernel<")
                bad += 0 if same else 1
                print("T%d-%d 5x5 s2 in %-18s skip %d: %s (%s, max diff %.2e)" % (cin, cout, shape, sk is not None,
                      "bit-identical" if same else "DIFFERS", name, (want - got).abs().max().item()))
    print("transposed 5x5 persistent form: %d mismatches" % bad)
    for (cin, cout), shape in PROD.items():
        w = (0.1 * torch.randn(cin, cout, 1, 5, 5, generator=g)).to(dev)
        layer = cp.ConvLayer(w, True, (1, 2, 2), (0, 2, 2), relu=False)
        x = torch.randn(*shape, cin, generator=g).to(dev)
        B, D, H, W = shape
        skip = torch.randn(B, D, 2 * H, 2 * W, cout, generator=g).to(dev)
        row = "T%d-%d %-18s" % (cin, cout, shape)
        for name, tiles in (("plan", None), ("direct", (1, 1, 0)), ("persistent", (2, 1, 5))):
            t = timeit(lambda: layer(x, skip=skip, skip_mode=cp.SKIP_ADD, tiles=tiles) if tiles else layer(x, skip=skip, skip_mode=cp.SKIP_ADD))
            row += "  %s %.1f us" % (name, t)
        print(row)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
