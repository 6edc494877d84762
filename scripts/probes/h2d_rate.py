#!/usr/bin/env python3
"""Pinned host -> device copy rate as bench.py's value_with_h2d loop issues it: five [1,3,512,640] fp32 images per depth
map (19.66 MB) with copy_(non_blocking=True) on one stream; also one 19.66 MB and one 256 MB buffer."""
import time

import torch

dev = torch.device("cuda:0")
st = torch.cuda.Stream(device=dev)


def rate(host, devt, reps=50):
    with torch.cuda.stream(st):
        for _ in range(5):
            for d, h in zip(devt, host):
                d.copy_(h, non_blocking=True)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for d, h in zip(devt, host):
                d.copy_(h, non_blocking=True)
        st.synchronize()
        el = time.perf_counter() - t0
    mb = sum(h.numel() for h in host) * 4 / 1e6
    return mb * reps / el / 1e3, 1e3 * el / reps


for label, shapes in (("5 x [1,3,512,640]", [(1, 3, 512, 640)] * 5), ("1 x 19.66 MB", [(5, 3, 512, 640)]),
                      ("1 x 256 MB", [(64, 1024, 1024)])):
    host = [torch.randn(*s).pin_memory() for s in shapes]
    devt = [torch.empty(*s, device=dev) for s in shapes]
    gbps, ms = rate(host, devt, reps=20 if "256" in label else 50)
    print("%-20s %6.2f GB/s  %7.3f ms per set" % (label, gbps, ms))
