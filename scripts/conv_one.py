#!/usr/bin/env python3
"""Run one convolution layer shape repeatedly with a given kernel choice (for rocprofv3 counter passes).
    conv_one.py CIN COUT KD KHW STRIDE B D H W MT NT VARIANT [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402

cin, cout, kd, k, st, B, D, H, W, mt, nt, var = (int(v) for v in sys.argv[1:13])
reps = int(sys.argv[13]) if len(sys.argv) > 13 else 20
dev = torch.device("cuda:0")
w = torch.randn(cout, cin, kd, k, k, device=dev) * 0.1
layer = cp.ConvLayer(w, False, (1, st, st), (kd // 2, k // 2, k // 2), relu=True)
x = torch.randn(B, D, H, W, cin, device=dev)
for _ in range(reps):
    layer(x, tiles=(mt, nt, var))
torch.cuda.synchronize()
