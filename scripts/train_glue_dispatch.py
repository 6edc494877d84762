#!/usr/bin/env python3
"""Who asks for the torch (aten) kernels of one training step: every aten op that reaches the device, with the innermost
frame of this package on the Python stack (forward and custom backward functions) or the autograd node it runs under
(built-in backward nodes), counted per (op, origin, shapes).  Eager launches; config 4 on one rank."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

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

VIEW_OPS = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "detach", "alias",
            "as_strided", "t.default", "_unsafe_view", "unbind", "split", "empty", "size", "stride", "is_", "sym_", "_local_scalar",
            "lift_fresh", "prim", "narrow", "unfold", "zeros_like", "ones_like", "new_empty", "new_zeros", "result_type",
            "_version", "set_", "record_stream", "chunk", "resolve_", "contiguous", "_to_copy")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.phase = "fwd"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(v in name for v in VIEW_OPS) and "copy" not in name:
            return out
        origin = None
        for fr in reversed(traceback.extract_stack()):
            if "/mvster_amd/" in fr.filename:
                origin = "%s:%d %s" % (fr.filename.split("/mvster_amd/")[-1], fr.lineno, fr.name)
                break
        if origin is None:
            node = torch._C._current_autograd_node()
            origin = "node:" + (type(node).__name__ if node is not None else "-")
        shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
        self.rows[(self.phase, name, origin, str(shapes)[:70])] += 1
        return out


def step(log=None):
    opt.zero_grad()
    if log:
        log.phase = "fwd"
    out = model(imgs, proj, dv)
    if log:
        log.phase = "loss"
    loss = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1,
                        ot_continous=False, mono=True)[0]
    if log:
        log.phase = "bwd"
    loss.backward()
    if log:
        log.phase = "opt"
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
log = Log()
with log:
    step(log)
torch.cuda.synchronize()
tot = sum(log.rows.values())
print("aten ops reaching the device in one step: %d" % tot)
by_origin = collections.Counter()
for (ph, name, origin, shp), n in log.rows.items():
    by_origin[(ph, origin)] += n
print("---- by origin")
for (ph, origin), n in by_origin.most_common(60):
    print("%4d  %-5s %s" % (n, ph, origin))
print("---- detail")
for (ph, name, origin, shp), n in sorted(log.rows.items(), key=lambda kv: (kv[0][0], kv[0][2], kv[0][1])):
    print("%4d  %-5s %-34s %-50s %s" % (n, ph, name.replace("aten.", ""), origin, shp))
