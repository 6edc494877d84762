#!/usr/bin/env python3
"""Roofline table of the training-only kernels of one step (config 4: 512x640, 5 views, B=2): every weight-gradient call
(slot kernel + finish) and every warp/aggregation backward, re-run in isolation on the tensors of a real step and timed
inside a hipGraph.  FLOPs / bytes are algorithmic: 2 * voxels * taps * CO * CI for a weight gradient (MFMA roof 157.3
TFLOP/s fp32), and for the warp backward the bytes it must move once -- reference and source features read, both
gradients written, hypotheses / forward outputs / output gradients read (HBM roof 8 TB/s).  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import SHIPPED, load_weights  # noqa: E402
from conv_microbench import timeit  # noqa: E402
from mvster_amd import MVS4net, MVS4net_loss, ops  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
H, W, N, B = 512, 640, 5, 2
model = MVS4net(**SHIPPED)
model.load_state_dict(load_weights(), strict=True)
if "--coherent" in sys.argv:
    with torch.no_grad():
        for r in model.reg:
            r.prob.weight.zero_()
model.to(dev).train()
imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev, batch=B)
g = torch.Generator().manual_seed(0)
gt, mask = {}, {}
for s in range(1, 5):
    hs, ws = H // 2 ** (4 - s), W // 2 ** (4 - s)
    gt["stage%d" % s] = (500 + 300 * torch.rand(B, hs, ws, generator=g)).to(dev)
    mask["stage%d" % s] = (torch.rand(B, hs, ws, generator=g) > 0.2).float().to(dev)

wg_calls, bwd_calls = [], []
orig_wg, orig_bwd = ops.conv_wgrad, ops.warp_agg_bwd_cl


def rec_wg(x, gy, kernel, stride, padding, **kw):
    wg_calls.append((x.detach().clone(), gy.detach().clone(), kernel, stride, padding, kw))
    return orig_wg(x, gy, kernel, stride, padding, **kw)


def rec_bwd(*a, **kw):
    bwd_calls.append(([t.detach().clone() if torch.is_tensor(t) else t for t in a], kw))
    return orig_bwd(*a, **kw)


ops.conv_wgrad, ops.warp_agg_bwd_cl = rec_wg, rec_bwd
out = model(imgs, proj, dv)
loss = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1,
                    ot_continous=False, mono=True)[0]
loss.backward()
torch.cuda.synchronize()
ops.conv_wgrad, ops.warp_agg_bwd_cl = orig_wg, orig_bwd

print("weight gradients (slot kernel + finish), fp32 MFMA roof 157.3 TFLOP/s")
tot = 0.0
rows = []
for x, gy, kernel, stride, padding, kw in wg_calls:
    us = min(timeit(lambda: orig_wg(x, gy, kernel, stride, padding, **kw), n=6) for _ in range(2))
    vox = gy.numel() // gy.shape[-1]
    fl = 2.0 * vox * kernel[0] * kernel[1] * kernel[2] * x.shape[-1] * gy.shape[-1]
    nbytes = 4.0 * (x.numel() + gy.numel())
    rows.append((us, "x %-22s gy %-22s k%s s%s" % ("x".join(map(str, x.shape)), "x".join(map(str, gy.shape)),
                                                   "x".join(map(str, kernel)), "x".join(map(str, stride))), fl, nbytes))
    tot += us
for us, desc, fl, nbytes in sorted(rows, key=lambda r: -r[0]):
    print("  %8.1f us  %6.2f TFLOP/s  frac %.3f   %7.0f GB/s   %s" % (us, fl / us / 1e6, fl / us / 1e6 / 157.3, nbytes / us / 1e3, desc))
print("  sum %.1f us over %d calls" % (tot, len(rows)))

print("warp/aggregation backward, HBM roof 8000 GB/s")
for a, kw in bwd_calls:
    ref, src, rt, hypo, fo, wsum, go = a[:7]
    us = min(timeit(lambda: orig_bwd(*a, **kw), n=4) for _ in range(2))
    nbytes = 4.0 * (2 * ref.numel() + 2 * src.numel() + hypo.numel() + fo.numel() + wsum.numel() + go.numel())
    print("  %8.1f us  %7.0f GB/s  frac %.3f   ref %-18s src %-22s D=%d" % (
        us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000.0, "x".join(map(str, ref.shape)), "x".join(map(str, src.shape)),
        hypo.shape[1]))
