#!/usr/bin/env python3
"""Pair form of the 3x3x3 Winograd layers (conv_wino_pair_kernel: variant word 9 | 3 << 8, nt = 1): bit-identical to the ring
kernel's mode 0 on ragged and production shapes (even depth), and timing against it and against the plan's choice.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from conv_microbench import timeit  # noqa: E402
from conv_wino_check import make_layer  # noqa: E402

dev = torch.device("cuda:0")
PAIR = 9 | (3 << 8)
SHAPES = [(1, 4, 38, 70), (2, 2, 9, 40), (1, 8, 16, 20), (1, 6, 8, 32), (3, 4, 13, 63)]
PROD = {16: [(1, 4, 256, 320), (1, 4, 128, 160), (1, 8, 64, 80), (1, 8, 32, 40)],
        32: [(1, 4, 128, 160), (1, 4, 64, 80), (1, 8, 32, 40), (1, 8, 16, 20)]}


def main():
    bad = 0
    for c in (16, 32):
        layer, _ = make_layer(c, c, 3)
        for shape in SHAPES + PROD[c]:
            x = torch.randn(*shape, c, device=dev)
            skip = torch.randn(*shape, c, device=dev)
            for sk in (None, skip):
                sm = 0 if sk is None else 1
                ref = layer(x, skip=sk, skip_mode=sm, tiles=(2, 1, 9))
                got = layer(x, skip=sk, skip_mode=sm, tiles=(2, 1, PAIR))
                torch.cuda.synchronize()
                same = torch.equal(ref, got)
                bad += 0 if same else 1
                print("%d->%d 3x3x3 in %-18s skip %d: %s (max diff %.2e)" % (c, c, shape, sm, "bit-identical" if same else "DIFFERS",
                                                                               (ref - got).abs().max().item()))
    print("pair form: %d mismatches" % bad)
    for c in (16, 32):
        layer, _ = make_layer(c, c, 3)
        for shape in PROD[c]:
            x = torch.randn(*shape, c, device=dev)
            row = "%d->%d %-18s" % (c, c, shape)
            for name, tiles in (("plan", None), ("ring nt1", (2, 1, 9)), ("pair", (2, 1, PAIR))) + \
                    ((("ring m2", (2, 2, 9 | (1 << 8))),) if c == 32 else ()):
                t = timeit(lambda: layer(x, tiles=tiles) if tiles else layer(x))
                row += "  %s %.1f us" % (name, t)
            print(row)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
