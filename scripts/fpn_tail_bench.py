"""Isolated timings of the FPN fine-level pieces (re-associated plan vs the direct layers) at the headline size."""
import sys
import torch
sys.path.insert(0, ".")
from mvster_amd import conv_plan as cp, modules as M, ops
from mvster_amd.conv_plan import SKIP_ADD, SKIP_UPSAMPLE_ADD

dev = "cuda"
NB, H, W = 5, 512, 640
torch.manual_seed(0)
m = M.FPN4(base_channels=8).to(dev).eval()
plan = cp.FpnPlan(m)
inner2, out3 = cp._plain2d(m.inner2), cp._plain2d(m.out3)
tail_g_half = plan.tail_g
c0 = torch.randn(NB, 1, H, W, 8, device=dev)
c1 = torch.randn(NB, 1, H // 2, W // 2, 16, device=dev)
f1 = torch.randn(NB, 1, H // 4, W // 4, 64, device=dev)


def timeit(name, fn, reps=30):
    best = None
    for _ in range(3):                 # best of three rounds (the first round of a new shape also pays the allocator)
        for _ in range(5):
            out = fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            out = fn()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / reps * 1e3
        best = us if best is None else min(best, us)
    print("%-44s %8.1f us" % (name, best), flush=True)
    return out


f2 = timeit("old  inner2(c1)+up(f1) -> f2", lambda: inner2(c1, skip=f1, skip_mode=SKIP_UPSAMPLE_ADD))
timeit("old  out3(f2)", lambda: out3(f2))
g4o = timeit("old  tail_g(f2) @ half res", lambda: tail_g_half(f2))
g3 = timeit("new  mid_g(f1) 64->144 @ quarter", lambda: plan.mid_g(f1))
for t in ((4, 3), (2, 3), (1, 3), (2, 9), (1, 9), (4, 1)):
    timeit("       mid_g tiles %s" % (t,), lambda: plan.mid_g(f1, tiles=t))
p3 = timeit("new  gather16 -> p3", lambda: ops.fpn_tail_gather(g3, plan.mid_vb, H // 2, W // 2))
timeit("new  mid_c(c1)+p3 -> o3", lambda: plan.mid_c(c1, skip=p3, skip_mode=SKIP_ADD))
g41 = timeit("new  tail_g(f1) @ quarter", lambda: plan.tail_g(f1))
g4 = timeit("new  lateral_up -> g4", lambda: ops.fpn_lateral_up(c1, plan.tail_a, plan.tail_ab, g41))
p4 = timeit("both gather8 -> p4", lambda: ops.fpn_tail_gather(g4, plan.tail_vb, H, W))
timeit("both tail_c(c0)+p4 -> o4", lambda: plan.tail_c(c0, skip=p4, skip_mode=SKIP_ADD))
timeit("new  whole tail", lambda: plan.tail(c0, c1, f1))
