import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mvster_amd import ops
from mvster_amd.synthetic import make_inputs
dev = torch.device("cuda:0")
H, W, N, B = 512, 640, 5, 2
_, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=0, batch=B)
lo, hi = dv[0, 0].item(), dv[0, -1].item()
C, G, D, s = 8, 4, 4, 3
g = torch.Generator().manual_seed(0)
ref = torch.randn(B, H, W, C, generator=g).to(dev); src = torch.randn(N - 1, B, H, W, C, generator=g).to(dev)
rt = ops.relative_projection(proj["stage4"].to(dev))
span = (1 / lo - 1 / hi) / 441.0
for noise in (0.0, 0.05, 0.15, 0.5, 1.0):
    c = 1 / hi + (1 / lo - 1 / hi) * (0.1 + 0.8 * ((1 - noise) * 0.5 + noise * torch.rand(B, 1, H, W, generator=g)))
    hypo = (1.0 / (c + span * (torch.arange(D).view(1, D, 1, 1) - 1.5))).float().contiguous().to(dev)
    out, wsum = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, want_wsum=True)
    gout = torch.randn_like(out)
    for _ in range(2): ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0)
    e1.record(); torch.cuda.synchronize()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record()
    for _ in range(5): ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0)
    f1.record(); torch.cuda.synchronize()
    print("stage-4 shape, per-pixel depth noise %.2f of the range: bwd %.3f ms  fwd %.3f ms" % (noise, e0.elapsed_time(e1) / 5, f0.elapsed_time(f1) / 5), flush=True)
