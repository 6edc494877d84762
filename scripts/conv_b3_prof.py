#!/usr/bin/env python3
"""A few launches of the bf16-split convolution kernel (conv_b3.hip) on the layers it targets, for the PMC passes
(MVSTER_LIB=...libmvster_hip_probes.so scripts/gpu_pmc_script.sh "scripts/conv_b3_prof.py" conv_b3,conv_wino)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402

dev = torch.device("cuda:0")
CASES = [(16, 16, 1, (5, 1, 256, 320), 1, 2), (16, 16, 1, (5, 1, 256, 320), 2, 1), (32, 32, 1, (5, 1, 128, 160), 2, 1),
         (64, 64, 1, (5, 1, 64, 80), 2, 1), (64, 64, 3, (1, 8, 8, 10), 1, 1), (64, 64, 3, (1, 8, 8, 10), 2, 1),
         (32, 32, 3, (1, 4, 128, 160), 1, 1)]
for cin, cout, kd, (B, D, H, W), tyq, wpc in CASES:
    g = torch.Generator().manual_seed(1)
    w = torch.randn(cout, cin, kd, 3, 3, generator=g) * 0.05
    layer = cp.ConvLayer(w.to(dev), False, (1, 1, 1), (kd // 2, 1, 1), relu=True)
    x = torch.randn(B, D, H, W, cin, generator=g).to(dev)
    for _ in range(4):
        layer(x, tiles=(tyq, 1, 11 | (wpc << 8)))
    for _ in range(2):
        layer(x)                      # today's kernel of the same layer, for comparison
    torch.cuda.synchronize()
