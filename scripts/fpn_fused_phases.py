#!/usr/bin/env python3
"""Finest FPN level at 5 x 512 x 640, isolated hipGraph timings: the fused launch against lateral + gather.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvster_amd import ops  # noqa: E402
from scripts.conv_microbench import timeit  # noqa: E402

dev = "cuda:0"
for NB, H, W in ((5, 512, 640),) if len(sys.argv) > 1 else ((5, 512, 640), (5, 1152, 1600), (7, 1024, 1920), (5, 832, 1152)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(NB, 1, H // 2, W // 2, 16, generator=g).to(dev)
    A = (torch.randn(72, 16, generator=g) * 0.3).to(dev)
    b = torch.randn(72, generator=g).to(dev)
    q = torch.randn(NB, 1, H // 4, W // 4, 72, generator=g).to(dev)
    vb = torch.randn(9, 8, generator=g).to(dev)
    G4 = ops.fpn_lateral_up(x, A, b, q)
    t_lat = min(timeit(lambda: ops.fpn_lateral_up(x, A, b, q), n=10) for _ in range(3))
    t_gat = min(timeit(lambda: ops.fpn_tail_gather(G4, vb, H, W), n=10) for _ in range(3))
    t_fus = min(timeit(lambda: ops.fpn_tail_fused(x, A, b, q, vb, H, W), n=10) for _ in range(3))
    mb = (x.numel() + q.numel() + NB * H * W * 8) * 4 / 1e6
    print("%dx%dx%d: lateral %.1f us + gather %.1f us = %.1f us; fused %.1f us (%.1f MB algorithmic, %.2f TB/s)" % (
        NB, H, W, t_lat, t_gat, t_lat + t_gat, t_fus, mb, mb / t_fus), flush=True)
    del G4
