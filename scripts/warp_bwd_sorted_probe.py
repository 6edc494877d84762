#!/usr/bin/env python3
"""Warp backward at one cascade stage (B=2, 512x640, 5 views) with RANDOM-WINNER hypotheses (every pixel's hypotheses
anywhere in the range, unrelated to its neighbours'): the sorted scatter against the window / atomic form.
usage: warp_bwd_sorted_probe.py <stage 1..4> [reps]   (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from mvster_amd import ops  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402
from oracle import mvs4_oracle as O  # noqa: E402  (hypothesis schedule only)

dev = torch.device("cuda:0")
H, W, N, B = 512, 640, 5, 2
s = int(sys.argv[1]) - 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
C, G, D = ((64, 8, 8), (32, 8, 8), (16, 4, 4), (8, 4, 4))[s]
_, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=0, batch=B)
h, w = H >> (3 - s), W >> (3 - s)
g = torch.Generator().manual_seed(s)
ref = torch.randn(B, h, w, C, generator=g).to(dev)
src = torch.randn(N - 1, B, h, w, C, generator=g).to(dev)
rt = ops.relative_projection(proj["stage%d" % (s + 1)].to(dev))
nfull = 8 * (1, 7, 21, 63)[s]
full = O.init_inverse_range(dv, nfull, h, w)
pick = torch.randint(0, max(nfull - D, 1), (B, 1, h, w), generator=g)
hypo = torch.gather(full, 1, pick + torch.arange(D).view(1, D, 1, 1)).contiguous().to(dev)
out, wsum = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, want_wsum=True)
gout = torch.randn_like(out)


def timeit(fn):
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


only = os.environ.get("PROBE_ONLY")
res = {}
for name, kw in (("sorted", dict(sorted_scatter=True)), ("atomic", dict(sorted_scatter=False))):
    if only and only != name:
        continue
    res[name] = timeit(lambda: ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, **kw))
print("stage %d (C=%d D=%d %dx%d) random winners: %s" % (s + 1, C, D, h, w, "  ".join("%s %.3f ms" % kv for kv in res.items())), flush=True)
