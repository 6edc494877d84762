#!/usr/bin/env python3
"""aten-level view of one native training step (config 4, one rank): which torch ops surround the HIP kernels."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from bench import SHIPPED, load_weights  # noqa: E402
from mvster_amd import MVS4net, MVS4net_loss  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
H, W, N, B = 512, 640, 5, 2
model = MVS4net(**SHIPPED)
model.load_state_dict(load_weights(), strict=True)
model.to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev, batch=B)
g = torch.Generator().manual_seed(0)
gt, mask = {}, {}
for s in range(1, 5):
    hs, ws = H // 2 ** (4 - s), W // 2 ** (4 - s)
    gt["stage%d" % s] = (500 + 300 * torch.rand(B, hs, ws, generator=g)).to(dev)
    mask["stage%d" % s] = (torch.rand(B, hs, ws, generator=g) > 0.2).float().to(dev)


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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=400, max_name_column_width=70))
print(prof.key_averages().table(sort_by="count", row_limit=5, max_name_column_width=70))
