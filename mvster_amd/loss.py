"""Training losses with the reference's signatures.  The Sinkhorn term of ``MVS4net_loss`` / ``Blend_loss`` runs as
one fused kernel (``mvster_sinkhorn``: per-pixel loss + its gradient, no [B,HW,D,D] intermediates).

``MVS4net_loss`` / ``Blend_loss`` follow models/MVS4Net.py:113-206; the OT term is the loss value of the reference's
``sinkhorn`` (models/mvs4net_utils.py:1096-1142): an entropy-regularised optimal-transport distance between the one-hot
ground-truth depth bin and the predicted ``attn_weight`` distribution, solved by ``iters`` log-domain Sinkhorn
updates with the |i-j| bin-distance cost.  (The tensor-level restatement that also returns the transport plan is test
infrastructure and lives with the other checkers, not in this package.)
"""
import torch


class _SinkhornLoss(torch.autograd.Function):
    """Masked mean of the per-pixel OT loss; gradient flows to ``attn_weight`` only (like the reference, where the
    target and the cost are constants)."""

    @staticmethod
    def forward(ctx, attn, hypo, gt, mask, iters, eps, continuous):
        from . import ops
        m = mask.to(torch.float32).contiguous()
        loss_pix, jac = ops.sinkhorn_pixels(attn.contiguous(), hypo.contiguous(), gt.contiguous(), iters, eps, mask=m,
                                            continuous=continuous)
        n = m.sum()
        ctx.save_for_backward(jac, m, n)
        # masked mean without a boolean-index gather (which would synchronise); where() keeps a non-finite loss of a
        # masked-out pixel (ground-truth depth 0) out of the sum
        return torch.where(m > 0.5, loss_pix, 0.0).sum() / n

    @staticmethod
    def backward(ctx, g):
        jac, m, n = ctx.saved_tensors
        w = (m * (g / n)).unsqueeze(1)
        return torch.where(w != 0, jac * w, 0.0), None, None, None, None, None, None


def sinkhorn_loss(gt_depth, hypo_depth, attn_weight, mask, iters, eps=1, continuous=False):
    """The loss value of ``sinkhorn`` (its second return) on the fused gfx950 kernel (``mvster_sinkhorn`` /
    ``mvster_sinkhorn_continuous``): GPU tensors, D <= 16 hypotheses (what ``MVS4net`` accepts as ``stage_splits``; the shipped 4/8 keep a
    pixel's whole problem in registers, 9..16 spill the iteration history to scratch), iters <= 16.  There is no tensor-level fallback: other inputs raise."""
    D = attn_weight.shape[1]
    if not attn_weight.is_cuda:
        raise RuntimeError("mvster_amd.loss.sinkhorn_loss runs on MI355X only (there is no CPU fallback)")
    if not ((3 if continuous else 2) <= D <= 16 and 0 <= iters <= 16):
        raise NotImplementedError("sinkhorn_loss: D=%d hypotheses / %d iterations (the fused kernel holds a pixel's whole "
                                  "problem in one thread: D <= 16, iters <= 16)" % (D, iters))
    return _SinkhornLoss.apply(attn_weight, hypo_depth, gt_depth, mask, int(iters), float(eps), bool(continuous))


def _masked_mean(values, m, n=None):
    """``values[mask].mean()`` without the boolean-index gather (that gather needs the number of selected elements on the
    host, i.e. a device synchronisation in the middle of every training step); ``m`` = the mask as floats, ``n`` = its sum
    if the caller already has it."""
    return (values * m).sum() / (m.sum() if n is None else n)


def _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
    inverse = kwargs.get("inverse_depth", False)
    ot_iter = kwargs.get("ot_iter", 3)
    ot_eps = kwargs.get("ot_eps", 1)
    ot_continous = kwargs.get("ot_continous", False)
    mono = kwargs.get("mono", False)
    dev = mask_ms["stage1"].device
    for stage_idx, key in enumerate([k for k in inputs.keys() if "stage" in k]):
        st = inputs[key]
        hypo, attn = st["hypo_depth"], st["attn_weight"]
        mask = mask_ms[key] > 0.5
        gt = depth_gt_ms[key]
        # (one float mask and one pixel count per stage, one reciprocal of the hypotheses: the same values as the
        #  reference's expressions, a third of the small launches)
        m = mask.to(torch.float32)
        n = m.sum()
        if mono and stage_idx != 0:
            l1 = _masked_mean((st["mono_depth"] - gt).abs(), m, n)         # F.l1_loss(mono_depth[mask], gt[mask])
        else:
            l1 = torch.zeros((), dtype=torch.float32, device=dev)
        with torch.no_grad():                                               # (a diagnostic: no gradient flows through it)
            if inverse:
                inv = 1 / hypo
                itv = (inv[:, 2] - inv[:, 1]).abs()
                inside = ((inv - (1 / gt).unsqueeze(1)).abs() <= itv.unsqueeze(1)).any(1)
            else:
                itv = (hypo[:, 2] - hypo[:, 1]).abs()
                inside = ((hypo - gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)).any(1)
            outside_ratio = _masked_mean((~inside).to(torch.float32), m, n)
        ot = sinkhorn_loss(gt, hypo, attn, mask, iters=ot_iter, eps=ot_eps, continuous=ot_continous)
        yield stage_idx, key, l1, ot, outside_ratio, mask


def MVS4net_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    stage_lw = kwargs.get("stage_lw", [1, 1, 1, 1])
    l1ot_lw = kwargs.get("l1ot_lw", [0, 1])
    total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    l1s, ots, ranges = [], [], []
    for si, _, l1, ot, rng, _ in _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
        l1s.append(l1)
        ots.append(ot)
        ranges.append(rng)
        total = total + stage_lw[si] * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot)
    return total, l1s, ots, ranges


def Blend_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    stage_lw = kwargs.get("stage_lw", [1, 1, 1, 1])
    l1ot_lw = kwargs.get("l1ot_lw", [0, 1])
    depth_max = kwargs.get("depth_max", 100)
    depth_min = kwargs.get("depth_min", 1)
    total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    l1s, ots, ranges = [], [], []
    last = None
    for si, key, l1, ot, rng, mask in _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
        l1s.append(l1)
        ots.append(ot)
        ranges.append(rng)
        total = total + stage_lw[si] * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot)
        last = (key, mask)
    key, mask = last
    scale = 128 / (depth_max - depth_min)[:, None, None]
    err = torch.abs(inputs[key]["depth"] * scale - depth_gt_ms[key] * scale)[mask]
    return total, l1s, ots, ranges, err.mean(), (err <= 3).float().mean() * 100, (err <= 1).float().mean() * 100
