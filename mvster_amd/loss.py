"""Training losses with the reference's signatures.  The Sinkhorn term of ``MVS4net_loss`` / ``Blend_loss`` runs as
one fused kernel (``mvster_sinkhorn``: per-pixel loss + its gradient, no [B,HW,D,D] intermediates); the terms around it
(valid mask, monocular L1, out-of-range flag, masked means) as a second one plus a single reduction
(``mvster_stage_loss_terms``, ``stage_losses``).

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


class _StageLoss(torch.autograd.Function):
    """One stage of MVS4net_loss (models/MVS4Net.py:131-151) in three launches: the fused Sinkhorn kernel (per-pixel OT
    loss and its gradient), ``mvster_stage_loss_terms`` (the valid mask, |mono - gt|, the out-of-range flag and the masked
    OT loss per pixel) and one sum over the planes.  Returns (l1, ot, out_of_range_ratio): the masked means
    ``F.l1_loss(mono[mask], gt[mask])``, ``sinkhorn(...)[1]`` and ``mask_out_of_range[mask].float().mean()``.  Gradients
    flow to ``attn`` and ``mono``; the ratio is a diagnostic.  As tensor expressions these were ~30 launches per stage."""

    @staticmethod
    def forward(ctx, attn, mono, hypo, gt, mask, iters, eps, continuous, inverse):
        from . import ops
        m = mask.to(torch.float32).contiguous()
        attn, hypo, gt = attn.contiguous(), hypo.contiguous(), gt.contiguous()
        loss_pix, jac = ops.sinkhorn_pixels(attn, hypo, gt, iters, eps, mask=m, continuous=continuous)
        terms = ops.stage_loss_terms(hypo, gt, m, loss_pix, None if mono is None else mono.contiguous(), inverse)
        # valid pixels, sum |mono - gt|, out of range, sum OT loss.  (In two steps where the pixel count allows: a reduction
        #  to four numbers runs on four workgroups' worth of the chip -- 89 us for the 4 x 655 360 planes of stage 4.)
        n_pix = terms.shape[1]
        if n_pix % 512 == 0:
            sums = terms[:4].view(4, n_pix // 512, 512).sum(2).sum(1)
        else:
            sums = terms[:4].sum(1)
        means = sums[1:] / sums[0]
        ctx.save_for_backward(jac, terms, sums)
        ctx.has_mono = mono is not None
        ctx.set_materialize_grads(False)
        l1, ratio, ot = means.unbind(0)
        ctx.mark_non_differentiable(ratio)
        return l1, ot, ratio

    @staticmethod
    def backward(ctx, g_l1, g_ot, _g_ratio):
        jac, terms, sums = ctx.saved_tensors
        B, _, H, W = jac.shape
        g_attn = g_mono = None
        if g_ot is not None and ctx.needs_input_grad[0]:
            w = (terms[0] * (g_ot / sums[0])).view(B, 1, H, W)
            g_attn = torch.where(w != 0, jac * w, 0.0)   # (where(): a non-finite Jacobian of a masked-out pixel stays out)
        if g_l1 is not None and ctx.has_mono and ctx.needs_input_grad[1]:
            g_mono = (terms[4] * (g_l1 / sums[0])).view(B, H, W)
        return g_attn, g_mono, None, None, None, None, None, None, None


def stage_losses(gt_depth, hypo_depth, attn_weight, mask, mono_depth=None, iters=3, eps=1, continuous=False, inverse=False):
    """(l1, ot, out_of_range_ratio) of one stage, as MVS4net_loss forms them (models/MVS4Net.py:131-151), on the fused
    gfx950 kernels; ``l1`` is 0 when ``mono_depth`` is None.  GPU tensors, 3 <= D <= 16 hypotheses, iters <= 16: there is no
    tensor-level fallback, other inputs raise."""
    D = attn_weight.shape[1]
    if not attn_weight.is_cuda:
        raise RuntimeError("mvster_amd.loss.stage_losses runs on MI355X only (there is no CPU fallback)")
    if not (3 <= D <= 16 and 0 <= iters <= 16):
        raise NotImplementedError("stage_losses: D=%d hypotheses / %d iterations (the fused kernels take 3 <= D <= 16, "
                                  "iters <= 16)" % (D, iters))
    l1, ot, ratio = _StageLoss.apply(attn_weight, mono_depth, hypo_depth, gt_depth, mask, int(iters), float(eps),
                                     bool(continuous), bool(inverse))
    if mono_depth is None:
        l1 = torch.zeros((), dtype=torch.float32, device=attn_weight.device)
    return l1, ot, ratio


def _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
    inverse = kwargs.get("inverse_depth", False)
    ot_iter = kwargs.get("ot_iter", 3)
    ot_eps = kwargs.get("ot_eps", 1)
    ot_continous = kwargs.get("ot_continous", False)
    mono = kwargs.get("mono", False)
    for stage_idx, key in enumerate([k for k in inputs.keys() if "stage" in k]):
        st = inputs[key]
        l1, ot, outside_ratio = stage_losses(depth_gt_ms[key], st["hypo_depth"], st["attn_weight"], mask_ms[key],
                                             st["mono_depth"] if mono and stage_idx != 0 else None, iters=ot_iter,
                                             eps=ot_eps, continuous=ot_continous, inverse=inverse)
        yield stage_idx, key, l1, ot, outside_ratio


def MVS4net_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    stage_lw = kwargs.get("stage_lw", [1, 1, 1, 1])
    l1ot_lw = kwargs.get("l1ot_lw", [0, 1])
    total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    l1s, ots, ranges = [], [], []
    for si, _, l1, ot, rng in _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
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
    for si, key, l1, ot, rng in _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
        l1s.append(l1)
        ots.append(ot)
        ranges.append(rng)
        total = total + stage_lw[si] * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot)
        last = key
    key = last
    mask = mask_ms[key] > 0.5
    scale = 128 / (depth_max - depth_min)[:, None, None]
    err = torch.abs(inputs[key]["depth"] * scale - depth_gt_ms[key] * scale)[mask]
    return total, l1s, ots, ranges, err.mean(), (err <= 3).float().mean() * 100, (err <= 1).float().mean() * 100
