#!/usr/bin/env python3
"""Host-side (Python) cost of one eager training step: cProfile over a few steps, top functions by own time."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import SHIPPED, load_weights  # noqa: E402
from mvster_amd import MVS4net, MVS4net_loss  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
H, W, N, B = 512, 640, 5, 2
model = MVS4net(**SHIPPED)
model.load_state_dict(load_weights(), strict=True)
model.to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev, batch=B)
g = torch.Generator().manual_seed(0)
gt = {"stage%d" % s: (500 + 300 * torch.rand(B, H >> (4 - s), W >> (4 - s), generator=g)).to(dev) for s in range(1, 5)}
mask = {k: torch.ones_like(v) for k, v in gt.items()}


def step():
    opt.zero_grad()
    out = model(imgs, proj, dv)
    loss = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1,
                        ot_continous=False, mono=True)[0]
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
