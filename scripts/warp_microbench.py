#!/usr/bin/env python3
"""Time every launch form of the fused warp+correlation+aggregation kernel at the four cascade stages of a
given image size (hipGraph-captured launches, so the figure is kernel time, not host time)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from mvster_amd import ops  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=512)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--eager", action="store_true", help="plain launches, no hipGraph (for rocprofv3 counter passes)")
ap.add_argument("--variants", type=str, default="1,2,3,4")
ap.add_argument("--stages", type=str, default="1,2,3,4")
args = ap.parse_args()
dev = torch.device("cuda:0")
_, proj, dv = make_inputs(nviews=args.views, H=args.height, W=args.width, seed=0)
NV = args.views - 1
VARIANTS = tuple(int(v) for v in args.variants.split(","))
print("MVSTER_PIX_NW=%s" % os.environ.get("MVSTER_PIX_NW", "1"))
print("%-28s %s" % ("stage (C,G,D,h,w)", "  ".join("v%d us / GB/s" % v for v in VARIANTS)))
for s, (C, G, D) in enumerate(((64, 8, 8), (32, 8, 8), (16, 4, 4), (8, 4, 4))):
    if str(s + 1) not in args.stages.split(","):
        continue
    h, w = args.height >> (3 - s), args.width >> (3 - s)
    g = torch.Generator().manual_seed(s)
    ref = torch.randn(1, h, w, C, generator=g).to(dev)
    src = torch.randn(NV, 1, h, w, C, generator=g).to(dev)
    lo, hi = dv[0, 0].item(), dv[0, -1].item()
    hypo = (lo + (hi - lo) * torch.linspace(0.2, 0.8, D).view(1, D, 1, 1) + torch.rand(1, D, h, w, generator=g)).to(dev)
    rt = ops.relative_projection(proj["stage%d" % (s + 1)].to(dev))
    nbytes = 4 * (ref.numel() + src.numel() + hypo.numel() * (1 + G))
    cells = []
    for variant in VARIANTS:
        if variant == 4 and C > 32:
            cells.append("      -        ")
            continue
        for _ in range(3):
            ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, variant=variant)
        torch.cuda.synchronize()
        if args.eager:
            for _ in range(args.reps):
                ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, variant=variant)
            torch.cuda.synchronize()
            cells.append("   eager       ")
            continue
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(args.reps):
                ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, variant=variant)
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * args.reps)
        cells.append("%7.2f / %6.0f" % (us, nbytes / us / 1e3))
    print("%-28s %s" % ("s%d (%d,%d,%d,%d,%d)" % (s + 1, C, G, D, h, w), "  ".join(cells)))
