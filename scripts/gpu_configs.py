#!/usr/bin/env python3
"""BASELINE.json configs on one MI355X: the larger inference shapes (runnable forms of configs 3 and 5),
one DDP-style train step (config 4, one rank), and property checks.  Prints one JSON line per config."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import SHIPPED, load_weights  # noqa: E402
from mvster_amd import MVS4net, MVS4net_loss  # noqa: E402
from mvster_amd.graph import GraphedForward  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")


def infer(H, W, N, steps=20):
    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    model.to(dev).eval()
    imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev)
    out = model(imgs, proj, dv)
    torch.cuda.synchronize()
    ok = all(torch.isfinite(out["stage%d" % s]["depth"]).all().item() for s in range(1, 5))
    attn_ok = all(((out["stage%d" % s]["attn_weight"].sum(1) - 1).abs().max() < 1e-5).item() for s in range(1, 5))
    g = GraphedForward(model, imgs, proj, dv)
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    same = torch.equal(g.outputs["depth"], out["depth"])
    print(json.dumps({"config": "%dx%d N=%d 4-stage eval" % (H, W, N), "ms_per_depth_map": round(dt * 1e3, 3),
                      "depth_maps_per_s": round(1 / dt, 2), "finite": ok, "softmax_ok": attn_ok, "graph_equals_eager": same,
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)
    del g, model
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()


FUSED_ADAM = os.environ.get("MVSTER_UNFUSED_ADAM") is None     # torch.optim.Adam(fused=True): one multi-tensor kernel


def train(H, W, N, B, steps=5, graph=False, coherent=False):
    """``coherent``: zero the four ``prob`` heads, so that every pixel picks hypothesis 0 and the depth maps the cascade
    hands from stage to stage are smooth, as they are for a network that has trained for a while.  The fixture weights
    are random: their winner-take-all depths are unrelated between neighbouring pixels, which turns the scatter of the
    warp backward into one global atomic per (tap, channel) -- the arithmetic per element is the same in both cases."""
    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    if coherent:
        with torch.no_grad():
            for r in model.reg:
                r.prob.weight.zero_()
                r.prob.weight.requires_grad_(False)      # (an optimizer step would make the winners random again)
    model.to(dev).train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, fused=FUSED_ADAM)
    imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev, batch=B)
    g = torch.Generator().manual_seed(0)
    gt, mask = {}, {}
    for s in range(1, 5):
        hs, ws = H // 2 ** (4 - s), W // 2 ** (4 - s)
        gt["stage%d" % s] = (500 + 300 * torch.rand(B, hs, ws, generator=g)).to(dev)
        mask["stage%d" % s] = (torch.rand(B, hs, ws, generator=g) > 0.2).float().to(dev)
    losses = []
    if graph:
        from mvster_amd.graph import GraphedTrainStep
        if os.environ.get("MVSTER_TORCH_ADAM") is None:
            from mvster_amd.optim import FusedAdam
            opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=1e-3)
        else:
            opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, capturable=True, fused=FUSED_ADAM)

        def loss_fn(o, g_, m_):
            return MVS4net_loss(o, g_, m_, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1,
                                ot_continous=False, mono=True)
        step = GraphedTrainStep(model, opt, loss_fn, imgs, proj, dv, gt, mask, warmup=3)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ls = [step().clone() for _ in range(steps)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        losses = [l.item() for l in ls]
    else:
        for i in range(steps + 2):
            if i == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            opt.zero_grad()
            out = model(imgs, proj, dv)
            loss = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10,
                                ot_eps=1, ot_continous=False, mono=True)[0]
            loss.backward()
            opt.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": "train %dx%d N=%d B=%d (1 rank, %s, OT loss), gfx950 kernels, %s" % (
        H, W, N, B, "Adam(fused=True)" if FUSED_ADAM else "Adam", ("one hipGraph per step" if graph else "eager launches") + (", smooth depth maps (prob heads zeroed)" if coherent else
                                                                                ", random-weight (incoherent) depth maps")),
                      "s_per_step": round(dt, 4), "loss_first": round(losses[0], 4), "loss_last": round(losses[-1], 4),
                      "finite": all(l == l for l in losses),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}), flush=True)


if __name__ == "__main__":
    infer(512, 640, 5)
    infer(1152, 1600, 5, steps=10)
    infer(1024, 1920, 7, steps=10)
    train(512, 640, 5, 2, steps=5)
