import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvster_amd import ops
from mvster_amd.synthetic import make_inputs
dev = torch.device("cuda:0")
_, proj, dv = make_inputs(nviews=5, H=512, W=640, seed=0, batch=2)
for s, (C, G, D) in enumerate(((64, 8, 8), (32, 8, 8), (16, 4, 4), (8, 4, 4))):
    h, w = 512 >> (3 - s), 640 >> (3 - s)
    g = torch.Generator().manual_seed(s)
    ref = torch.randn(2, h, w, C, generator=g).to(dev); src = torch.randn(4, 2, h, w, C, generator=g).to(dev)
    lo, hi = dv[0, 0].item(), dv[0, -1].item()
    hypo = (lo + (hi - lo) * torch.linspace(0.3, 0.7, D).view(1, D, 1, 1) + torch.rand(2, D, h, w, generator=g)).to(dev)
    rt = ops.relative_projection(proj["stage%d" % (s + 1)].to(dev))
    out, wsum = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, want_wsum=True)
    go = torch.randn_like(out)
    for _ in range(2): ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, go, G, True, True, 2.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, go, G, True, True, 2.0)
    e1.record(); torch.cuda.synchronize()
    print("stage %d bwd (C=%d) %.2f ms" % (s + 1, C, e0.elapsed_time(e1) / 5), flush=True)
