#!/usr/bin/env python3
"""Stage-4 warp kernel with warm, cold and pre-touched inputs: does its in-forward time (48-51 us against 30-32 us in the
back-to-back microbenchmark) come from where the features are?  cold = a 1 GB fill between launches (evicts L2 and the
256 MB memory-side cache); touched = cold, then a streaming read of the features on the same stream before the launch."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from mvster_amd import ops  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
_, proj, dv = make_inputs(nviews=5, H=512, W=640, seed=0)
for s, (C, G, D) in ((3, (8, 4, 4)), (2, (16, 4, 4)), (1, (32, 8, 8))):
    h, w = 512 >> (3 - s), 640 >> (3 - s)
    g = torch.Generator().manual_seed(s)
    ref = torch.randn(1, h, w, C, generator=g).to(dev)
    src = torch.randn(4, 1, h, w, C, generator=g).to(dev)
    lo, hi = dv[0, 0].item(), dv[0, -1].item()
    hypo = (lo + (hi - lo) * torch.linspace(0.2, 0.8, D).view(1, D, 1, 1) + torch.rand(1, D, h, w, generator=g)).to(dev)
    rt = ops.relative_projection(proj["stage%d" % (s + 1)].to(dev))
    big = torch.empty(256 << 20, device=dev)              # 1 GB

    def run(mode, reps=20):
        tot = 0.0
        for _ in range(reps):
            if mode != "warm":
                big.fill_(1.0)
            if mode == "touched":
                src.sum(); ref.sum(); hypo.sum()
            torch.cuda._sleep(2_000_000)                    # the launch is queued before the GPU gets to it
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps * 1e3
    for _ in range(3):
        ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0)
    print("stage %d (C=%d): warm %.1f us, cold %.1f us, cold + touched %.1f us" % (s + 1, C, run("warm"), run("cold"), run("touched")))
