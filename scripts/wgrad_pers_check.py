#!/usr/bin/env python3
"""Persistent weight-gradient kernel (conv_wgrad_pers.hip) against the staged kernels: run once per mode
    python scripts/wgrad_pers_check.py pers      /     MVSTER_WGRAD_NO_PERS=1 python scripts/wgrad_pers_check.py staged
(the switch is read when the library loads), then `python scripts/wgrad_pers_check.py compare`.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
CASES = [  # (CI, CO, kd, x shape [B, D, H, W])
    (16, 16, 1, (10, 1, 256, 320)), (32, 32, 1, (10, 1, 128, 160)), (64, 64, 1, (10, 1, 64, 80)), (64, 32, 1, (10, 1, 128, 160)),
    (16, 16, 3, (2, 4, 256, 320)), (32, 32, 3, (2, 4, 128, 160)), (64, 64, 3, (2, 4, 64, 80)), (16, 16, 3, (2, 8, 64, 80)),
    (32, 32, 3, (2, 8, 32, 40)), (16, 16, 1, (3, 1, 37, 70)), (32, 16, 3, (1, 3, 9, 130)), (16, 32, 1, (2, 2, 5, 64)),
]
mode = sys.argv[1]
if mode == "compare":
    a, b = torch.load(os.path.join(OUT, "wgrad_pers.pt")), torch.load(os.path.join(OUT, "wgrad_staged.pt"))
    bad = 0
    for (case, da, ta), (_, db, tb) in zip(a, b):
        err = (da - db).abs().max().item() / db.abs().max().item()
        ok = err < 2e-5
        bad += 0 if ok else 1
        print("%-34s pers %7.1f us  staged %7.1f us  (x%.2f)  max |d| / max |dW| %.2e %s" % (case, ta, tb, tb / ta, err, "" if ok else "BAD"))
    print("sum: pers %.1f us, staged %.1f us" % (sum(t for _, _, t in a), sum(t for _, _, t in b)))
    sys.exit(1 if bad else 0)
from conv_microbench import timeit  # noqa: E402
from mvster_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
res = []
for ci, co, kd, shape in CASES:
    g = torch.Generator().manual_seed(ci + co + kd + shape[2])
    x = torch.randn(*shape, ci, generator=g).to(dev)
    gy = torch.randn(*shape, co, generator=g).to(dev)
    k, p = (kd, 3, 3), (kd // 2, 1, 1)
    dw = ops.conv_wgrad(x, gy, k, (1, 1, 1), p)
    name = _lib.last_kernel()
    us = min(timeit(lambda: ops.conv_wgrad(x, gy, k, (1, 1, 1), p), n=5) for _ in range(2))
    case = "%d->%d k%d %s" % (ci, co, kd, "x".join(map(str, shape)))
    print("%-34s %8.1f us  %s" % (case, us, name), flush=True)
    res.append((case, dw.cpu(), us))
torch.save(res, os.path.join(OUT, "wgrad_%s.pt" % mode))
