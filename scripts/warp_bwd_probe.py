#!/usr/bin/env python3
"""Warp backward in isolation at the four cascade stages (B=2, 512x640, 5 views): kernel time with smooth and with noisy
hypothesis maps, gather vs atomic window flush, and how reproducible the source gradient is."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from mvster_amd import ops  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
H, W, N, B = 512, 640, 5, 2
_, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=0, batch=B)
lo, hi = dv[0, 0].item(), dv[0, -1].item()


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for s, (C, G, D) in enumerate(((64, 8, 8), (32, 8, 8), (16, 4, 4), (8, 4, 4))):
    h, w = H >> (3 - s), W >> (3 - s)
    g = torch.Generator().manual_seed(s)
    ref = torch.randn(B, h, w, C, generator=g).to(dev)
    src = torch.randn(N - 1, B, h, w, C, generator=g).to(dev)
    rt = ops.relative_projection(proj["stage%d" % (s + 1)].to(dev))
    # inverse-depth hypotheses around a smooth surface, spacing like the cascade's stage s
    span = (1 / lo - 1 / hi) / (7.0 * (7.0 if s >= 1 else 1) * (3.0 if s >= 2 else 1) * (3.0 if s >= 3 else 1))
    yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
    centre = 1 / hi + (1 / lo - 1 / hi) * (0.3 + 0.4 * xx * yy)
    for label, noise in (("smooth", 0.0), ("noisy", 0.15)):
        c = centre + noise * (1 / lo - 1 / hi) * torch.rand(B, 1, h, w, generator=g)
        hypo = (1.0 / (c + span * (torch.arange(D).view(1, D, 1, 1) - (D - 1) / 2))).float().contiguous().to(dev)
        out, wsum = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, want_wsum=True)
        gout = torch.randn_like(out)
        res = {}
        for det in (True, False):
            res[det] = timeit(lambda: ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, deterministic=det))
        a = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, deterministic=True)
        b = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, deterministic=True)
        diff = (a[1] != b[1]).float().mean().item()
        fwd = timeit(lambda: ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0))
        print("stage %d (C=%d D=%d %dx%d) %-6s  bwd gather %7.3f ms  atomic %7.3f ms  fwd %6.3f ms  g_src elements differing between two "
              "gather runs: %.2e" % (s + 1, C, D, h, w, label, res[True], res[False], fwd, diff), flush=True)
