"""Training losses with the reference's signatures.  The Sinkhorn term of ``MVS4net_loss`` / ``Blend_loss`` runs as
one fused kernel (``mvster_sinkhorn``: per-pixel loss + its gradient, no [B,HW,D,D] intermediates); the terms around it
(valid mask, monocular L1, out-of-range flag, masked means, the weighted sum over the stages) as ``mvster_stage_loss_fwd``
forward and ``mvster_stage_loss_bwd`` backward (``_StageLoss``).

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
    """One stage of MVS4net_loss (models/MVS4Net.py:131-151) in three launches forward -- the fused Sinkhorn kernel (per-pixel
    OT loss and its gradient), ``mvster_stage_loss_fwd`` (valid mask, |mono - gt|, out-of-range flag, masked OT loss: the
    per-workgroup sums, then the means and the stage's weighted term added to the running total) -- and ONE backward
    (``mvster_stage_loss_bwd``).  Returns (l1, ot, out_of_range_ratio, total): the masked means
    ``F.l1_loss(mono[mask], gt[mask])``, ``sinkhorn(...)[1]``, ``mask_out_of_range[mask].float().mean()`` and
    ``total_in + stage_lw * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot)``.  Gradients flow to ``attn``, ``mono`` and ``total_in``;
    the ratio is a diagnostic.  As tensor expressions these were ~30 launches per stage forward and as many backward."""

    @staticmethod
    def forward(ctx, attn, mono, total_in, hypo, gt, mask, iters, eps, continuous, inverse, w_l1, w_ot, w_stage):
        from . import ops
        m = mask.to(torch.float32).contiguous()
        attn, hypo, gt = attn.contiguous(), hypo.contiguous(), gt.contiguous()
        loss_pix, jac = ops.sinkhorn_pixels(attn, hypo, gt, iters, eps, mask=m, continuous=continuous)
        planes, out = ops.stage_loss_fwd(hypo, gt, m, loss_pix, None if mono is None else mono.contiguous(),
                                         None if total_in is None else total_in.detach().reshape(1).contiguous(), inverse,
                                         w_l1, w_ot, w_stage)
        ctx.save_for_backward(jac, planes, out)
        ctx.has_mono = mono is not None
        ctx.weights = (w_stage * w_l1, w_stage * w_ot)
        ctx.set_materialize_grads(False)
        _n, l1, ratio, ot, _weighted, total = out.unbind(0)
        ctx.mark_non_differentiable(ratio)
        return l1, ot, ratio, total

    @staticmethod
    def backward(ctx, g_l1, g_ot, _g_ratio, g_total):
        from . import ops
        jac, planes, out = ctx.saved_tensors
        want_attn = ctx.needs_input_grad[0] and (g_ot is not None or g_total is not None)
        want_mono = ctx.has_mono and ctx.needs_input_grad[1] and (g_l1 is not None or g_total is not None)
        g_attn = g_mono = None
        if want_attn or want_mono:
            def scalar(g):
                return None if g is None else g.reshape(1).to(torch.float32).contiguous()
            g_attn, g_mono = ops.stage_loss_bwd(jac, planes, out, scalar(g_total), scalar(g_l1), scalar(g_ot), ctx.weights[0],
                                                ctx.weights[1], want_attn, want_mono)
        return (g_attn, g_mono, g_total if ctx.needs_input_grad[2] else None) + (None,) * 10


def _check_stage(attn_weight, iters, name):
    D = attn_weight.shape[1]
    if not attn_weight.is_cuda:
        raise RuntimeError("mvster_amd.loss.%s runs on MI355X only (there is no CPU fallback)" % name)
    if not (3 <= D <= 16 and 0 <= iters <= 16):
        raise NotImplementedError("%s: D=%d hypotheses / %d iterations (the fused kernels take 3 <= D <= 16, "
                                  "iters <= 16)" % (name, D, iters))


def stage_losses(gt_depth, hypo_depth, attn_weight, mask, mono_depth=None, iters=3, eps=1, continuous=False, inverse=False):
    """(l1, ot, out_of_range_ratio) of one stage, as MVS4net_loss forms them (models/MVS4Net.py:131-151), on the fused
    gfx950 kernels; ``l1`` is 0 when ``mono_depth`` is None.  GPU tensors, 3 <= D <= 16 hypotheses, iters <= 16: there is no
    tensor-level fallback, other inputs raise."""
    _check_stage(attn_weight, iters, "stage_losses")
    l1, ot, ratio, _ = _StageLoss.apply(attn_weight, mono_depth, None, hypo_depth, gt_depth, mask, int(iters), float(eps),
                                        bool(continuous), bool(inverse), 1.0, 1.0, 1.0)
    return l1, ot, ratio


def stage_losses_total(gt_depth, hypo_depth, attn_weight, mask, mono_depth, total, iters, eps, continuous, inverse, w_l1, w_ot,
                       w_stage):
    """``stage_losses`` plus the running total of MVS4net_loss: -> (l1, ot, out_of_range_ratio, total + w_stage * (w_l1 * l1 +
    w_ot * ot)) (models/MVS4Net.py:131-151), ``total`` None for the first stage.  One fused forward, one fused backward."""
    _check_stage(attn_weight, iters, "MVS4net_loss")
    return _StageLoss.apply(attn_weight, mono_depth, total, hypo_depth, gt_depth, mask, int(iters), float(eps), bool(continuous),
                            bool(inverse), float(w_l1), float(w_ot), float(w_stage))


def _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
    """Per stage: (index, key, l1, ot, out_of_range_ratio, running total) -- the total of MVS4Net.py:151 is carried through
    the stages' fused kernels instead of ~5 scalar launches per stage."""
    inverse = kwargs.get("inverse_depth", False)
    ot_iter = kwargs.get("ot_iter", 3)
    ot_eps = kwargs.get("ot_eps", 1)
    ot_continous = kwargs.get("ot_continous", False)
    mono = kwargs.get("mono", False)
    stage_lw = kwargs.get("stage_lw", [1, 1, 1, 1])
    l1ot_lw = kwargs.get("l1ot_lw", [0, 1])
    total = None
    for stage_idx, key in enumerate([k for k in inputs.keys() if "stage" in k]):
        st = inputs[key]
        l1, ot, outside_ratio, total = stage_losses_total(
            depth_gt_ms[key], st["hypo_depth"], st["attn_weight"], mask_ms[key], st["mono_depth"] if mono and stage_idx != 0 else None,
            total, ot_iter, ot_eps, ot_continous, inverse, l1ot_lw[0], l1ot_lw[1], stage_lw[stage_idx])
        yield stage_idx, key, l1, ot, outside_ratio, total


def MVS4net_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    total = None
    l1s, ots, ranges = [], [], []
    for _si, _, l1, ot, rng, total in _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
        l1s.append(l1)
        ots.append(ot)
        ranges.append(rng)
    if total is None:
        total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    return total, l1s, ots, ranges


def Blend_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    depth_max = kwargs.get("depth_max", 100)
    depth_min = kwargs.get("depth_min", 1)
    total = None
    l1s, ots, ranges = [], [], []
    last = None
    for _si, key, l1, ot, rng, total in _stage_terms(inputs, depth_gt_ms, mask_ms, kwargs):
        l1s.append(l1)
        ots.append(ot)
        ranges.append(rng)
        last = key
    if total is None:
        total = torch.zeros((), dtype=torch.float32, device=mask_ms["stage1"].device)
    key = last
    mask = mask_ms[key] > 0.5
    scale = 128 / (depth_max - depth_min)[:, None, None]
    err = torch.abs(inputs[key]["depth"] * scale - depth_gt_ms[key] * scale)[mask]
    return total, l1s, ots, ranges, err.mean(), (err <= 3).float().mean() * 100, (err <= 1).float().mean() * 100
