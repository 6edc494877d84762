#!/usr/bin/env python3
"""Probe: fp32 products on the bf16 matrix cores (3-way split operands, six MFMAs, fp32 accumulate; conv_b3.hip, variant 11)
against the kernels the plan runs today, layer by layer: max error against an fp64 convolution (in units of max |y|) and
hipGraph-timed duration.  GPU only, probe library:
`MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_probes.so python scripts/conv_b3_check.py [quick]`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from conv_microbench import timeit  # noqa: E402
from mvster_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
# (cin, cout, kd, (B, D, H, W), skip)   -- the layers of the 512x640x5 forward the Winograd kernels carry today
LAYERS = [
    (16, 16, 1, (5, 1, 256, 320), False), (16, 16, 1, (5, 1, 256, 320), True),
    (32, 32, 1, (5, 1, 128, 160), False), (64, 64, 1, (5, 1, 64, 80), False), (64, 32, 1, (5, 1, 128, 160), False),
    (16, 16, 3, (1, 4, 256, 320), False), (32, 32, 3, (1, 4, 128, 160), False), (64, 64, 3, (1, 4, 64, 80), False),
    (16, 16, 3, (1, 4, 128, 160), False), (32, 32, 3, (1, 4, 64, 80), False), (64, 64, 3, (1, 4, 32, 40), False),
    (16, 16, 3, (1, 8, 64, 80), False), (32, 32, 3, (1, 8, 32, 40), False), (64, 64, 3, (1, 8, 16, 20), False),
    (16, 16, 3, (1, 8, 32, 40), False), (32, 32, 3, (1, 8, 16, 20), False), (64, 64, 3, (1, 8, 8, 10), False),
    # ragged sizes (tile edges in x and y, odd depth)
    (32, 32, 1, (2, 1, 37, 50), True), (16, 16, 3, (1, 3, 21, 45), False), (64, 32, 1, (1, 1, 9, 33), False),
]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    LAYERS = LAYERS[:3] + LAYERS[-3:]
names = {0: "direct", 1: "lds", 2: "splitk", 5: "persistent", 8: "winograd", 9: "winograd-ring", 11: "bf16x3"}
tot_now = tot_b3 = tot_best = 0.0
worst_ratio = 0.0
for cin, cout, kd, (B, D, H, W), with_skip in LAYERS:
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + kd + H)
    w = torch.randn(cout, cin, kd, 3, 3, generator=g) * (2.0 / (cin * 9 * kd)) ** 0.5
    bn = torch.nn.BatchNorm3d(cout)
    bn.weight.data = 0.5 + torch.rand(cout, generator=g)
    bn.bias.data = torch.randn(cout, generator=g) * 0.1
    bn.running_mean.data = torch.randn(cout, generator=g) * 0.1
    bn.running_var.data = 0.5 + torch.rand(cout, generator=g)
    bn.eval()
    layer = cp.ConvLayer(w.to(dev), False, (1, 1, 1), (kd // 2, 1, 1), bn=bn.to(dev), relu=True)
    x = torch.randn(B, D, H, W, cin, generator=g).to(dev)
    skip = torch.randn(B, D, H, W, cout, generator=g).to(dev) if with_skip else None
    sm = cp.SKIP_ADD if with_skip else cp.SKIP_NONE
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.to(dev).double(), padding=(kd // 2, 1, 1))
    ref = ref * layer.scale[:cout].double().view(1, -1, 1, 1, 1) + layer.shift[:cout].double().view(1, -1, 1, 1, 1)
    ref = ref.clamp_min(0).permute(0, 2, 3, 4, 1)
    if with_skip:
        ref = ref + skip.double()
    scale = ref.abs().max().item()
    now = layer(x, skip=skip, skip_mode=sm)
    k_now = _lib.last_kernel()
    e_now = (now.double() - ref).abs().max().item() / scale
    t_now = min(timeit(lambda: layer(x, skip=skip, skip_mode=sm), n=10) for _ in range(2))
    # the direct fp32 MFMA kernel (no Winograd): the accuracy yardstick of a plain fp32 accumulation
    direct = layer(x, skip=skip, skip_mode=sm, tiles=(1, 1, 0))
    e_dir = (direct.double() - ref).abs().max().item() / scale
    res = []
    for tyq, wpc, prio in ((2, 1, 1), (1, 1, 1), (1, 1, 0), (1, 1, 2)):
        word = 11 | ((wpc | (prio << 2)) << 8)
        got = layer(x, skip=skip, skip_mode=sm, tiles=(tyq, 1, word))
        k_b3 = _lib.last_kernel()
        e = (got.double() - ref).abs().max().item() / scale
        t = min(timeit(lambda: layer(x, skip=skip, skip_mode=sm, tiles=(tyq, 1, word)), n=10) for _ in range(2))
        res.append((t, e, "TY%d/w%d/p%d" % (4 * tyq, wpc, prio)))
    best = min(res)
    fl = layer.flops(B, D, H, W)
    big = B * D * H * W >= 20000
    if big:
        tot_now += t_now
        tot_b3 += best[0]
        tot_best += min(best[0], t_now)
    worst_ratio = max(worst_ratio, max(r[1] for r in res) / max(e_dir, 1e-12))
    print("C%d-%d k%dx3x3 %dx%dx%dx%d%s | now %-38s %6.1f us %6.1f TF/s err %.1e | direct fp32 err %.1e | bf16x3 %s err %.1e -> best %s x%.2f (%.1f TF/s)" % (
        cin, cout, kd, B, D, H, W, " +skip" if with_skip else "", k_now, t_now, fl / t_now / 1e6, e_now, e_dir,
        "  ".join("%s %6.1f us" % (r[2], r[0]) for r in res), max(r[1] for r in res), best[2], t_now / best[0], fl / best[0] / 1e6), flush=True)
print("layers of >= 20 000 voxels: %.1f us today, %.1f us on the bf16-split kernel, %.1f us taking the faster per layer; worst error "
      "ratio bf16x3 / direct fp32: %.2f" % (tot_now, tot_b3, tot_best, worst_ratio))
