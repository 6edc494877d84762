"""Training losses with the reference's signatures.  On the GPU the Sinkhorn term of ``MVS4net_loss`` /
``Blend_loss`` runs as one fused kernel (``mvster_sinkhorn``: per-pixel loss + its gradient, no [B,HW,D,D]
intermediates); ``sinkhorn`` itself is the tensor-level form with the reference's return values.

``MVS4net_loss`` / ``Blend_loss`` follow models/MVS4Net.py:113-206, ``sinkhorn`` follows
models/mvs4net_utils.py:1096-1142: an entropy-regularised optimal-transport distance between
the one-hot ground-truth depth bin and the predicted ``attn_weight`` distribution, solved by
``iters`` log-domain Sinkhorn updates with the |i-j| bin-distance cost.
"""
import torch
import torch.nn.functional as F


def sinkhorn(gt_depth, hypo_depth, attn_weight, mask, iters, eps=1, continuous=False):
    B, D, H, W = attn_weight.shape
    dev = gt_depth.device
    pred = attn_weight.permute(0, 2, 3, 1).reshape(B, H * W, D)
    if not continuous:
        ar = torch.arange(D, dtype=torch.float32, device=dev)
        cost = (ar[None, :] - ar[:, None]).abs()[None, None].repeat(B, H * W, 1, 1)           # [B,HW,D,D]
        nearest = (hypo_depth - gt_depth[:, None]).abs().min(1)[1].reshape(B * H * W, 1)
        target = torch.zeros(B * H * W, D, dtype=hypo_depth.dtype, device=dev)
        target.scatter_add_(1, nearest, torch.ones(B * H * W, 1, dtype=hypo_depth.dtype, device=dev))
        target = target.reshape(B, H * W, D)
    else:
        target = torch.zeros((B, H * W, D + 1), dtype=torch.float32, device=dev)
        target[:, :, -1] = 1
        ar = torch.arange(D, dtype=torch.float32, device=dev)
        cost = torch.zeros((B, D, D + 1), dtype=torch.float32, device=dev)
        cost[:, :D, :D] = (ar[None, :] - ar[:, None]).abs()[None]
        cost = cost[:, None, None].repeat(1, H, W, 1, 1)
        itv = 1 / hypo_depth[:, 2] - 1 / hypo_depth[:, 1]
        off = (1 / gt_depth - 1 / hypo_depth[:, 0]) / itv
        off[~mask] = 10
        cost[..., -1] = torch.stack([(off - i).abs() for i in range(D)], dim=1).permute(0, 2, 3, 1)
        cost = cost.reshape(B, H * W, D, D + 1)
    log_mu = (target + 1e-12).log()
    log_nu = (pred + 1e-12).log()
    u, v = torch.zeros_like(log_nu), torch.zeros_like(log_mu)
    for _ in range(iters):
        v = log_mu - torch.logsumexp(cost / eps + u.unsqueeze(3), dim=2)
        u = log_nu - torch.logsumexp(cost / eps + v.unsqueeze(2), dim=3)
    plan = (cost / eps + u.unsqueeze(3) + v.unsqueeze(2)).exp()
    loss = (plan * cost).reshape(B * H * W, -1)[mask.reshape(-1)].sum(-1).mean()
    return plan, loss


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
        return torch.where(m > 0.5, loss_pix, torch.zeros_like(loss_pix)).sum() / n

    @staticmethod
    def backward(ctx, g):
        jac, m, n = ctx.saved_tensors
        w = (m * (g / n)).unsqueeze(1)
        return torch.where(w != 0, jac * w, torch.zeros_like(jac)), None, None, None, None, None, None


def sinkhorn_loss(gt_depth, hypo_depth, attn_weight, mask, iters, eps=1, continuous=False):
    """The loss value of ``sinkhorn`` (its second return) on the fused gfx950 kernel (``mvster_sinkhorn`` /
    ``mvster_sinkhorn_continuous``): GPU tensors, D <= 8 hypotheses, iters <= 16 -- the range of the reference's
    configurations.  There is no tensor-level fallback: other inputs raise (``sinkhorn`` above is the explicit
    reference-shaped form that also returns the transport plan)."""
    D = attn_weight.shape[1]
    if not attn_weight.is_cuda:
        raise RuntimeError("mvster_amd.loss.sinkhorn_loss runs on MI355X only (there is no CPU fallback)")
    if not ((3 if continuous else 2) <= D <= 8 and 0 <= iters <= 16):
        raise NotImplementedError("sinkhorn_loss: D=%d hypotheses / %d iterations (the fused kernel holds a pixel's whole "
                                  "problem in registers: D <= 8, iters <= 16)" % (D, iters))
    return _SinkhornLoss.apply(attn_weight, hypo_depth, gt_depth, mask, int(iters), float(eps), bool(continuous))


def _masked_mean(values, mask):
    """``values[mask].mean()`` without the boolean-index gather: that gather needs the number of selected elements
    on the host, i.e. a device synchronisation in the middle of every training step."""
    m = mask.to(values.dtype)
    return (values * m).sum() / m.sum()


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
        if mono and stage_idx != 0:
            l1 = _masked_mean((st["mono_depth"] - gt).abs(), mask)         # F.l1_loss(mono_depth[mask], gt[mask])
        else:
            l1 = torch.zeros((), dtype=torch.float32, device=dev)
        if inverse:
            itv = (1 / hypo[:, 2] - 1 / hypo[:, 1]).abs()
            outside = ((1 / hypo - 1 / gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)).sum(1) == 0
        else:
            itv = (hypo[:, 2] - hypo[:, 1]).abs()
            outside = ((hypo - gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)).sum(1) == 0
        ot = sinkhorn_loss(gt, hypo, attn, mask, iters=ot_iter, eps=ot_eps, continuous=ot_continous)
        yield stage_idx, key, l1, ot, _masked_mean(outside.float(), mask), mask


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
