#!/usr/bin/env python3
"""Per-layer table of one eval forward: every ConvLayer call with the kernel the plan picked for it, its
hipGraph-timed duration, achieved TFLOP/s and algorithmic GB/s.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from bench import SHIPPED, load_weights  # noqa: E402
from conv_microbench import timeit  # noqa: E402
from mvster_amd import MVS4net  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

H, W, N = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 640, 5)))
dev = torch.device("cuda:0")
model = MVS4net(**SHIPPED)
model.load_state_dict(load_weights(), strict=True)
model.to(dev).eval()
model.graph_cache = False          # instrumented eager launches (MVS4net.forward would replay a captured graph)
imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev)
calls = []
orig = cp.ConvLayer.__call__


def rec(layer, x, skip=None, skip_mode=0, tiles=None):
    calls.append((layer, tuple(x.shape), None if skip is None else tuple(skip.shape), skip_mode if skip is not None else 0))
    return orig(layer, x, skip, skip_mode, tiles)


cp.ConvLayer.__call__ = rec
model(imgs, proj, dv)
cp.ConvLayer.__call__ = orig
torch.cuda.synchronize()
names = {0: "direct", 1: "lds", 2: "splitk", 3: "small", 4: "deconv_small", 5: "persistent", 6: "persistent1x1", 7: "pingpong", 8: "winograd", 9: "winograd-ring", 10: "narrow-mfma"}
total = 0.0
for i, (layer, xs, ss, sm) in enumerate(calls):
    x = torch.randn(*xs, device=dev)
    skip = torch.randn(*ss, device=dev) if ss else None
    B, Di, Hi, Wi, _ = xs
    _, mt, nt, oshape, var = layer._geom(B, Di, Hi, Wi, sm)
    us = min(timeit(lambda: layer(x, skip=skip, skip_mode=sm), n=10) for _ in range(2))
    out_ch = 1 if layer.prob is not None else layer.cout
    nbytes = 4 * (x.numel() + (skip.numel() if skip is not None else 0) + oshape[0] * oshape[1] * oshape[2] * oshape[3] * out_ch)
    fl = layer.flops(B, Di, Hi, Wi)
    total += us
    print("%2d %-46s %-12s mt%d nt%d %8.1f us %7.2f TF/s %7.0f GB/s%s" % (
        i, cp.layer_signature(layer, B, Di, Hi, Wi, sm), names[var & 0xff], mt, nt, us, fl / us / 1e6, nbytes / us / 1e3,
        "  +prob" if layer.prob is not None else ""), flush=True)
print("sum of conv layers: %.1f us" % total)
