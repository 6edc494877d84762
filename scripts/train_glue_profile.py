#!/usr/bin/env python3
"""Which torch (aten) kernels surround the HIP kernels in one training step: count and GPU time per
(aten op, input shapes).  Eager launches; config 4 on one rank."""
import collections
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
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
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
stacks = "--stacks" in sys.argv
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=stacks) as prof:
    step()
    torch.cuda.synchronize()
if stacks:
    # --stacks: who asks for the small kernels -- (aten op, innermost frame inside this package or the autograd engine)
    src = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
            continue
        frame = "(autograd engine / no python frame)"
        for fr in (ev.stack or []):
            if "mvster_amd/" in fr or "bench.py" in fr or "scripts/" in fr:
                frame = fr.split("mvster_amd/")[-1] if "mvster_amd/" in fr else fr.split("/")[-1]
                break
        shp = str([tuple(x) for x in ev.input_shapes if x])[:60] if ev.self_device_time_total > 8 else ""
        a = src[(ev.name, frame[:100] + "  " + shp)]
        a[0] += 1
        a[1] += ev.self_device_time_total
    print("aten kernels by (op, calling line[, shapes of the large ones]), by time: %.2f ms in %d launches"
          % (sum(v[1] for v in src.values()) / 1e3, sum(v[0] for v in src.values())))
    for (name, frame), (n, t) in sorted(src.items(), key=lambda kv: -kv[1][1])[:70]:
        print("%5d x %8.1f us  %-24s %s" % (n, t, name, frame))
    sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    where = str([tuple(x) for x in ev.input_shapes if x])
    a = agg[(ev.name, where[:90])]
    a[0] += 1
    a[1] += ev.self_device_time_total
tot = sum(v[1] for v in agg.values())
print("aten kernels: %d launches, %.2f ms" % (sum(v[0] for v in agg.values()), tot / 1e3))
by_count = "--by-count" in sys.argv
only = [a for a in sys.argv[1:] if not a.startswith("--")]          # optional: aten op names to list in full
if by_count:
    names = collections.defaultdict(lambda: [0, 0.0])
    for (name, where), (n, t) in agg.items():
        names[name][0] += n
        names[name][1] += t
    for name, (n, t) in sorted(names.items(), key=lambda kv: -kv[1][0]):
        print("%5d x %8.1f us  %s" % (n, t, name))
for (name, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:(1000 if only else 110)]:
    if only and name.split("::")[1] not in only:
        continue
    print("%7.1f us %4d x  %-28s %s" % (t, n, name, where))
