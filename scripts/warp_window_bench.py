#!/usr/bin/env python3
"""Wave-local warp kernel (variant 3) against the LDS-staged source-window form (variant 5) at the two fine stages of a
512x640x5 depth map: smooth hypotheses (neighbouring pixels at similar depths: every tap is served from the staged
window) and per-pixel random hypotheses (the lane-by-lane fallback), with warm inputs (the launch repeated in a
hipGraph) and cold ones (a 600 MB fill between launches evicts L2 and the Infinity Cache: the in-forward situation)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from mvster_amd import ops  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
H, W, N = 512, 640, 5
_, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=0)
NV = N - 1
evict = torch.empty(150_000_000, device=dev)


def warm(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cold(fn, reps=8):
    ts = []
    for _ in range(reps):
        evict.fill_(1.0)
        torch.cuda._sleep(2_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


print("%-10s %-22s %-8s | %s" % ("stage", "(C,G,D,h,w)", "regime", "wave warm / cold us  |  window warm / cold us  (GB/s of the algorithmic bytes, cold)"))
for s, (C, G, D) in ((2, (16, 4, 4)), (3, (8, 4, 4))):
    h, w = H >> (3 - s), W >> (3 - s)
    g = torch.Generator().manual_seed(s)
    ref = torch.randn(1, h, w, C, generator=g).to(dev)
    src = torch.randn(NV, 1, h, w, C, generator=g).to(dev)
    lo, hi = dv[0, 0].item(), dv[0, -1].item()
    yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
    xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
    step = (hi - lo) / (64 if s == 3 else 32) * torch.arange(D).view(1, D, 1, 1)
    regimes = {
        "smooth": lo + (hi - lo) * (0.2 + 0.5 * xx + 0.2 * yy) + step + 0.5 * torch.rand(1, D, h, w, generator=g),
        "random": lo + (hi - lo) * torch.rand(1, D, h, w, generator=g),
    }
    rt = ops.relative_projection(proj["stage%d" % (s + 1)].to(dev))
    nbytes = 4 * (ref.numel() + src.numel() + h * w * D * (1 + G))
    for name, hypo in regimes.items():
        hypo = hypo.expand(1, D, h, w).contiguous().to(dev)
        base = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, variant=3)
        got = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, variant=5)
        same = torch.equal(base, got)
        cells = []
        for variant in (3, 5):
            fn = lambda: ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, variant=variant)   # noqa: E731
            cells.append((warm(fn), cold(fn)))
        print("stage %d    %-22s %-8s | %6.1f / %6.1f  |  %6.1f / %6.1f   (%4.0f vs %4.0f GB/s)  %s" % (
            s + 1, (C, G, D, h, w), name, cells[0][0], cells[0][1], cells[1][0], cells[1][1], nbytes / cells[0][1] / 1e3,
            nbytes / cells[1][1] / 1e3, "bit-identical" if same else "DIFFERENT"), flush=True)
