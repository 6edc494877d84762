"""``MVS4net`` -- the reference's model API on hand-written gfx950 kernels.

Drop-in for ``models.MVS4Net.MVS4net`` (reference models/MVS4Net.py:9-111): same constructor
keywords (including the ``depth_interals_ratio`` spelling), same
``forward(imgs, proj_matrices, depth_values, filename=None)``, same output dict (``stage1..4``
sub-dicts plus the last stage flattened at top level) and the same ``state_dict`` keys.

Execution:
* eval  -- every op of the cascade runs in libmvster_hip.so: FPN4 and the regularisation U-Nets on
  the fp32 MFMA convolution kernel (BatchNorm folded), one fused warp+correlation+attention launch
  per stage, fused prob+softmax+argmax+gather selection, hypothesis schedulers, confidence
  upsampling.  No host synchronisation inside ``forward`` (the reference's ``.cpu().numpy()``
  at MVS4Net.py:61-62 is gone), so the whole forward can be captured in a HIP graph
  (``mvster_amd.graph.GraphedForward``).
* train -- BatchNorm needs batch statistics and cannot be folded: the layers run conv -> BatchNorm
  -> ReLU on channels-last tensors with all three convolution passes (forward, input gradient,
  weight gradient) on the gfx950 kernels (``mvster_amd.train_ops``), and the fused
  warp/aggregation kernel is an ``autograd.Function`` with a hand-written HIP backward.
There is no CPU path and no PyTorch-ROCm / MIOpen path: CPU tensors raise, and the on-GPU cross-check of the
training step is the oracle module tree moved to the GPU (tests/test_gpu_train.py).
"""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .conv_plan import FpnPlan, Reg2dPlan, Reg3dPlan
from .modules import FPN4, mono_depth_decoder, reg2d, reg3d


class _WarpAggPyr(torch.autograd.Function):
    """cor_feats [B,D,h,w,G] from one level of the view-major channels-last pyramid [N*B,1,h,w,C] (the first B maps are the
    reference view's); HIP forward + backward.  Takes the level whole and returns ONE gradient for it, the kernel writing
    the two parts in place: slicing outside would have autograd zero-fill, copy and add a full-size buffer per slice on the
    way back."""

    @staticmethod
    def forward(ctx, pyr, B, rt, hypo, G, group_cor, attn_fuse_d, attn_temp):
        pyr = pyr.contiguous()
        ref_cl, src_cl = _WarpAggPyr._parts(pyr, B)
        out, wsum = ops.warp_agg_fwd_cl(ref_cl, src_cl, rt, hypo, G, group_cor, attn_fuse_d, attn_temp, want_wsum=True)
        ctx.save_for_backward(pyr, rt, hypo, out, wsum)
        ctx.cfg = (B, G, group_cor, attn_fuse_d, attn_temp)
        ctx.set_materialize_grads(False)
        # second output: the reference view's maps [B,1,h,w,C] for other consumers (the monocular head) -- their gradient
        # comes back here and is added to the B maps it belongs to, instead of a zero-filled full-size buffer per use
        return out, pyr[:B]

    @staticmethod
    def _parts(pyr, B):
        n, _, h, w, C = pyr.shape
        return pyr[:B].view(B, h, w, C), pyr[B:].view(n // B - 1, B, h, w, C)

    @staticmethod
    def backward(ctx, grad, grad_ref_maps):
        pyr, rt, hypo, out, wsum = ctx.saved_tensors
        B, G, group_cor, attn_fuse_d, attn_temp = ctx.cfg
        if grad is None:
            g_pyr = torch.zeros_like(pyr)
        else:
            g_pyr = torch.empty_like(pyr)
            g_ref, g_src = _WarpAggPyr._parts(g_pyr, B)
            ref_cl, src_cl = _WarpAggPyr._parts(pyr, B)
            n_src, _, hs, ws, c_ = src_cl.shape
            if not (ops.SORTED_SCATTER and ops.warp_agg_bwd_sorted_scratch(B, n_src, c_, hypo.shape[1], hypo.shape[2],
                                                                           hypo.shape[3], hs, ws) is not None):
                g_src.zero_()                                        # (window form: the source gradient is scattered with atomics)
            ops.warp_agg_bwd_cl(ref_cl, src_cl, rt, hypo, out, wsum, grad.contiguous(), G, group_cor, attn_fuse_d,
                                attn_temp, into=(g_ref, g_src))
        if grad_ref_maps is not None:
            g_pyr[:B].add_(grad_ref_maps)
        return g_pyr, None, None, None, None, None, None, None


class _SelectDepthCL(torch.autograd.Function):
    """reg2d's 1x1x1 ``prob`` head + softmax over depth + first-max argmax + gather + inverse bounds of one stage in ONE
    kernel each way (mvs4net_utils.py:900, :1068-1088).  Differentiable output: ``attn_weight`` (w.r.t. the 8-channel
    feature volume, ``prob.weight`` and ``prob.bias``); depth and the inverse bounds carry no gradient, as in the
    reference (argmax; the next stage detaches them)."""

    @staticmethod
    def forward(ctx, feat_cl, prob_w, prob_b, hypo, split_itv, inverse_depth):
        feat_cl = feat_cl.contiguous()
        w = prob_w.detach().reshape(-1).contiguous()
        sel = ops.select_depth(hypo, split_itv, inverse_depth, feat_cl=feat_cl, prob_w=w, prob_b=prob_b.detach().reshape(-1))
        ctx.save_for_backward(sel["attn_weight"], feat_cl, w)
        ctx.shapes = (prob_w.shape, prob_b.shape)
        ctx.set_materialize_grads(False)     # (no zero-filled gradients for the three non-differentiable outputs)
        outs = (sel["attn_weight"], sel["depth"]) + ((sel["inverse_min_depth"], sel["inverse_max_depth"]) if inverse_depth else ())
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, gattn, *unused):
        if gattn is None:
            return (None,) * 6
        attn, feat_cl, w = ctx.saved_tensors
        dfeat, dw, db = ops.select_depth_bwd(attn, gattn, feat_cl, w)
        return dfeat, dw.reshape(ctx.shapes[0]), db.reshape(ctx.shapes[1]), None, None, None


def _invalidate_after_load(module, incompatible):
    """load_state_dict() post hook (a module-level function: a lambda would make the module unpicklable)."""
    module.invalidate_plans()


class MVS4net(nn.Module):
    # hypotheses per stage: the fused forward kernels of the shipped cascade hold up to 16 per pixel in registers; beyond
    # that the forward runs on the general warp kernel (32 / 16 pixels per workgroup) and the memory-walking selection
    # kernel.  The backward kernels, the fused stage selection and the Sinkhorn kernel keep 16.
    MAX_HYPOTHESES = 64
    MAX_HYPOTHESES_TRAIN = 16
    _stream_warning_off = False

    def __init__(self, arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
                 stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=False,
                 group_cor_dim=[8, 8, 8, 8], inverse_depth=False, agg_type="ConvBnReLU3D", dcn=False, pos_enc=0,
                 mono=False, asff=False, attn_temp=2, attn_fuse_d=True, vis_ETA=False, vis_mono=False):
        super().__init__()
        if arch_mode != "fpn":
            raise NotImplementedError("arch_mode %r (only 'fpn' exists in the reference too)" % arch_mode)
        if dcn or asff or pos_enc or vis_ETA or vis_mono:
            raise NotImplementedError("dcn / asff / pos_enc / vis_* ablation switches are out of scope "
                                      "(SURVEY.md section 2, #10): not enabled by the shipped scripts")
        lo = 3 if inverse_depth else 2           # (the inverse-depth bounds read hypotheses 1 and 2, mvs4net_utils.py:1083)
        if max(stage_splits) > self.MAX_HYPOTHESES or min(stage_splits) < lo:
            raise NotImplementedError("stage_splits %r: the kernels take %d..%d depth hypotheses per stage in evaluation "
                                      "(3..%d in training; the shipped cascade uses 8/8/4/4)"
                                      % (stage_splits, lo, self.MAX_HYPOTHESES, self.MAX_HYPOTHESES_TRAIN))
        if not group_cor:
            # the squared-difference volume has one correlation per CHANNEL; the general warp kernel keeps them in LDS
            for s_, d_ in enumerate(stage_splits[:num_stage]):
                if d_ > 8 and fpn_base_channel * 2 ** (3 - s_) >= 32:
                    raise NotImplementedError("stage_splits %r with group_cor=False: stages with 32 or more feature channels "
                                              "take at most 8 hypotheses (stage %d has %d)" % (stage_splits, s_ + 1, d_))
        self.arch_mode = arch_mode
        self.num_stage = num_stage
        self.depth_interals_ratio = list(depth_interals_ratio)
        self.group_cor = group_cor
        self.group_cor_dim = list(group_cor_dim)
        self.inverse_depth = inverse_depth
        self.stage_splits = list(stage_splits)
        self.mono = mono
        self.attn_temp = attn_temp
        self.attn_fuse_d = attn_fuse_d
        self.reg_net = reg_net
        self.feature = FPN4(base_channels=fpn_base_channel)
        # empty lists kept for state_dict / attribute parity with the reference (MVS4Net.py:35,43)
        self.attn_ob = nn.ModuleList()
        self.pos_enc_func = nn.ModuleList()
        self.reg = nn.ModuleList()
        if self.mono:
            self.mono_depth_decoder = mono_depth_decoder()
        down_size = [3, 3, 2, 2]
        for idx in range(num_stage):
            in_dim = self.group_cor_dim[idx] if group_cor else self.feature.out_channels[idx]
            if reg_net == "reg2d":
                self.reg.append(reg2d(input_channel=in_dim, base_channel=reg_channel, conv_name=agg_type))
            elif reg_net == "reg3d":
                self.reg.append(reg3d(in_channels=in_dim, base_channels=reg_channel, down_size=down_size[idx]))
            else:
                raise NotImplementedError("reg_net %r" % reg_net)
        self._plans = {}               # device -> (FpnPlan, [RegPlan]); shared by nn.DataParallel replicas
        # run the fine FPN levels on a second HIP stream underneath cascade stages 1-2 (which are small,
        # latency-bound launches that leave most of the chip idle)
        self.overlap_streams = True
        # training: cascade stages whose forward AND backward run on ONE side stream.  The forward joins it before anything reads
        # a stage's outputs (its order is unchanged); autograd runs a node's backward on its forward's stream, and the stages'
        # backward passes are independent (a stage detaches what it takes from the previous one), so the three small stages'
        # latency-bound backward runs beside the fine stage's and the FPN's: 11.62 -> 10.99 ms per config-4 step (stages 1-2
        # only 11.33, each stage on its own stream 11.13, one of them alone: no gain).  () = everything on the caller's stream.
        self.train_side_stages = (0, 1, 2)
        self.train_side_separate = False   # each side stage on a stream of its own (measured slower)
        # training: the FPN's two fine levels (forward and backward) on their own stream, beside cascade stages 1-2
        self.train_fpn_tail_stream = True
        self._side_streams = {}
        self.warp_variant = 0          # mvster_warp_agg_fwd variant (0 = per-shape default)
        # hypothesis scheduling inside the warp launch (mvster_warp_agg_fwd_sched: bit-identical, one launch less per stage).
        # Off: measured 1 091 against 1 109 depth-maps/s and 1.131 against 1.123 ms for one forward alone (same box, alternating
        # runs, profiles/r05_fused_hypotheses_ab.txt) -- the warp kernels are VALU-bound and the ~35 instructions per lane cost
        # more than the 5 us scheduler launches, which co-run with the other depth map's kernels anyway
        self.fuse_hypotheses = False
        # pack + projections + first hypotheses in one launch, the three coarse confidence up-samplings in one (same bits)
        self.merge_launches = True
        # eval calls replay a captured hipGraph from the second call of a shape on (graph.ForwardCache); False = every
        # call issues its ~76 launches eagerly, as before round 5.  Shared by nn.DataParallel's single-device pass-through.
        self.graph_cache = True
        from .graph import ForwardCache
        self._fwd_cache = ForwardCache()
        self.register_load_state_dict_post_hook(_invalidate_after_load)

    def __getstate__(self):
        # copy.deepcopy(model) (EMA helpers) and torch.save(model): the caches hold device-side plans, HIP streams and
        # captured graphs -- none of it state; the copy rebuilds its own on first use
        from .graph import ForwardCache
        state = dict(self.__dict__)
        state["_plans"], state["_side_streams"], state["_fwd_cache"] = {}, {}, ForwardCache()
        return state

    # ------------------------------------------------------------------ plan cache
    def invalidate_plans(self):
        """Drop the packed-weight plans (eval) and the cached packed layers of the training path; they are
        rebuilt on the next forward.  Called automatically by load_state_dict(), by train()/eval() when the
        mode actually changes and by .to()/.cuda(); call it by hand after modifying parameters in place while
        in eval mode, or through ``p.data`` in training mode (see ``train_ops._LayerCache``)."""
        self._plans.clear()            # in place: nn.DataParallel replicas share the dict
        self._fwd_cache.clear()
        from . import train_ops
        train_ops.CACHE.clear()

    def train(self, mode=True):
        # the reference's validation loop calls .eval() for every sample (train_mvs4.py:258): only a real
        # mode change (parameters were trained in between) costs a plan rebuild
        if mode != self.training:
            self._plans.clear()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._plans.clear()
        self._fwd_cache.clear()        # (.to() / .cuda() / .float(): the captured graphs point at the old storage)
        return super()._apply(fn, *args, **kwargs)

    def _state_stamp(self):
        """Changes whenever a parameter or buffer the eval plans were folded from is written in place through the tensor
        (``p.mul_()``, ``copy_``, an eager optimizer step, an EMA swap: ``_version``), replaced (``p.data = t``: the storage
        address) or updated by a captured training step's replay (the per-parameter epoch cells of ``train_ops.CACHE``:
        a ``GraphedTrainStep`` bumps the cell of the parameters ITS optimizer owns, so another model in the process --
        a frozen teacher, an EMA copy -- keeps its plans).  348 Python attribute reads per forward, ~60 us; writes
        through a detached alias of the storage cannot be seen from here -- call ``invalidate_plans()`` after those."""
        import itertools
        from . import train_ops
        cells = train_ops.CACHE.cells
        # (running statistics written by a training-mode forward of the native BatchNorm: raw-pointer writes, no _version)
        acc = train_ops.CACHE.stat_writes * 15485863
        for t in itertools.chain(self.feature.parameters(), self.feature.buffers(), self.reg.parameters(), self.reg.buffers()):
            acc = (acc * 1000003 + t._version * 7919 + t.data_ptr()) & 0xFFFFFFFFFFFF
            if cells:
                c = cells.get(t)
                if c is not None:
                    acc += c[0] * 104729
        return acc

    def _get_plans(self):
        # per device: an nn.DataParallel replica on cuda:1 must not run plans whose packed weights live on cuda:0
        dev = next(self.parameters()).device
        hit = self._plans.get(dev)
        # (inside a graph capture the plans of the warm-up runs are the ones to record: no re-build there)
        stamp = hit[2] if hit is not None and torch.cuda.is_current_stream_capturing() else self._state_stamp()
        if hit is None or hit[2] != stamp:
            with torch.no_grad():
                fpn = FpnPlan(self.feature)
                regs = [Reg2dPlan(m) if isinstance(m, reg2d) else Reg3dPlan(m) for m in self.reg]
            hit = self._plans[dev] = (fpn, regs, stamp)
        return hit[0], hit[1]

    # ------------------------------------------------------------------ pieces
    def _hypotheses(self, stage_idx, depth_values, depth_interval, prev, H, W):
        D = self.stage_splits[stage_idx]
        if stage_idx == 0:
            return ops.init_range(depth_values, D, H, W, inverse=self.inverse_depth)
        if self.inverse_depth:
            return ops.schedule_inverse_range(prev["inverse_min_depth"].detach(), prev["inverse_max_depth"].detach(), D,
                                              H, W)
        return ops.schedule_range(prev["depth"].detach(), D, self.depth_interals_ratio[stage_idx] * depth_interval, H, W)

    def _check_inputs(self, imgs, proj_matrices, depth_values):
        if len(imgs) == 0 or not torch.is_tensor(imgs[0]):
            raise RuntimeError("imgs: a non-empty list of [B,3,H,W] tensors expected")
        if imgs[0].dim() != 4 or imgs[0].shape[1] != 3:
            raise RuntimeError("imgs: a list of [B,3,H,W] tensors expected, got %s" % (tuple(imgs[0].shape),))
        B, _, H, W = imgs[0].shape
        if H % 64 or W % 64:
            raise RuntimeError("image size %dx%d: H and W must be multiples of 64 (stage-1 is H/8 and reg2d "
                               "halves three more times)" % (H, W))
        N = len(imgs)
        for i, im in enumerate(imgs):
            if tuple(im.shape) != (B, 3, H, W) or im.device != imgs[0].device or im.dtype != torch.float32:
                raise RuntimeError("imgs[%d]: %s %s on %s, expected [%d,3,%d,%d] float32 (the path is fp32-only, like the "
                                   "reference's) on %s" % (i, tuple(im.shape), im.dtype, im.device, B, H, W, imgs[0].device))
        # the raw pointers of these go straight to the kernels: every stage's projection stack and the depth range are
        # checked here (reference layout: proj_matrices[stage] [B,N,2,4,4] = (extrinsic, intrinsic) per view,
        # datasets/dtu_yao4.py:176-189; depth_values [B,D>=2], MVS4Net.py:60-63)
        for s in range(self.num_stage):
            name = "stage%d" % (s + 1)
            if name not in proj_matrices:
                raise RuntimeError("proj_matrices has no %r entry (keys: %s)" % (name, sorted(proj_matrices.keys())))
            pm = proj_matrices[name]
            if tuple(pm.shape) != (B, N, 2, 4, 4):
                raise RuntimeError("proj_matrices[%r]: shape %s, expected [%d,%d,2,4,4] (batch, views = len(imgs))"
                                   % (name, tuple(pm.shape), B, N))
            if not pm.dtype.is_floating_point:
                raise RuntimeError("proj_matrices[%r]: floating-point tensor expected, got %s" % (name, pm.dtype))
        if N < 2:
            raise RuntimeError("at least one source view is needed (len(imgs) = %d)" % N)
        if depth_values.dim() != 2 or depth_values.shape[0] != B or depth_values.shape[1] < 2:
            raise RuntimeError("depth_values: shape %s, expected [%d, D >= 2] (the range is read from its first and last "
                               "column)" % (tuple(depth_values.shape), B))
        if not imgs[0].is_cuda:
            raise RuntimeError("mvster_amd.MVS4net runs on MI355X only: move the model and inputs to the GPU "
                               "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")

    # ------------------------------------------------------------------ eval: all-HIP
    @torch.no_grad()
    def _forward_eval(self, imgs, proj_matrices, depth_values, teacher=None, capture=None):
        N = len(imgs)
        B, _, H, W = imgs[0].shape
        dev = imgs[0].device
        fpn, regs = self._get_plans()
        depth_values = depth_values.to(dev, torch.float32)
        depth_interval = None
        if not self.inverse_depth:
            depth_interval = (depth_values[:, -1] - depth_values[:, 0]) / depth_values.size(1)
        names = ["stage%d" % (s + 1) for s in range(self.num_stage)]
        pms = [proj_matrices[n].to(dev, torch.float32) for n in names]
        hypo0 = None
        if self.merge_launches and (teacher is None or "stage1" not in teacher) and not self.fuse_hypotheses:
            # pack + projections + the first stage's hypotheses: one launch instead of three
            # (stage 1 reads the coarsest FPN level: H/8 x W/8)
            packed, rts, hypo0 = ops.forward_prologue(imgs, pms, depth_values, self.stage_splits[0], H // 8, W // 8,
                                                      self.inverse_depth)
        else:
            packed, rts = ops.pack_images(imgs), ops.relative_projection_multi(pms)
        c0, c1, c3, f1 = fpn.trunk(packed)                                       # channels-last [N*B,1,h,w,C]
        main = torch.cuda.current_stream()
        side = None
        if self.overlap_streams and self.num_stage > 2:
            side = self._side_streams.get(dev)
            if side is None:
                side = self._side_streams[dev] = torch.cuda.Stream(device=dev)
            side.wait_stream(main)                                               # fork: fine levels on `side`
            with torch.cuda.stream(side):
                o3, o4 = fpn.tail(c0, c1, f1)
            o1, o2 = fpn.coarse(c3, f1)
        else:
            o1, o2 = fpn.coarse(c3, f1)
            # a 1- or 2-stage cascade (BASELINE config 1) never reads the two fine levels
            o3, o4 = fpn.tail(c0, c1, f1) if self.num_stage > 2 else (None, None)
        pyramid = [o1, o2, o3, o4]

        outputs = {}
        prev = None
        for s in range(self.num_stage):
            name = "stage%d" % (s + 1)
            if s == 2 and side is not None:
                main.wait_stream(side)                                           # join: stage 3 reads o3, stage 4 o4
                if not torch.cuda.is_current_stream_capturing():
                    for t in (o3, o4):
                        t.record_stream(main)                                    # allocated on `side`, consumed on `main`
            f = pyramid[s]
            h, w, C = f.shape[2], f.shape[3], f.shape[4]
            f = f.view(N, B, h, w, C)
            ref_cl, src_cl = f[0], f[1:]
            G = self.group_cor_dim[s] if self.group_cor else C
            rt = rts[s]
            cor = hypo = None
            if teacher is not None and name in teacher:
                hypo = teacher[name].contiguous()
            elif self.inverse_depth and self.group_cor and self.warp_variant == 0 and self.fuse_hypotheses:
                # the stage's hypotheses are computed inside the warp launch (one kernel and one dependency edge less per stage)
                fused = ops.warp_agg_fwd_sched_cl(
                    ref_cl.contiguous(), src_cl.contiguous(), rt, G, self.stage_splits[s], self.attn_fuse_d, float(self.attn_temp),
                    inv_min=None if s == 0 else prev["inverse_min_depth"], inv_max=None if s == 0 else prev["inverse_max_depth"],
                    depth_values=depth_values.contiguous() if s == 0 else None)
                if fused is not None:
                    cor, hypo = fused
            if hypo is None and s == 0 and hypo0 is not None:
                hypo = hypo0
            if hypo is None:
                hypo = self._hypotheses(s, depth_values, depth_interval, prev, h, w)
            if cor is None:
                cor = ops.warp_agg_fwd_cl(ref_cl, src_cl, rt, hypo, G, self.group_cor, self.attn_fuse_d,
                                          float(self.attn_temp), variant=self.warp_variant)
            plan = regs[s]
            want_logits = capture is not None
            if isinstance(plan, Reg2dPlan):
                # (the U-Net's last layer, the prob head and the selection in one launch where the plan allows it)
                sel = plan.select(cor, hypo, self.depth_interals_ratio[s], self.inverse_depth, want_logits=want_logits)
            elif plan.fused_prob:
                sel = ops.select_depth(hypo, self.depth_interals_ratio[s], self.inverse_depth, feat_cl=plan(cor),
                                       prob_w=plan.prob_w, prob_b=plan.prob_b, want_logits=want_logits)
            else:
                sel = ops.select_depth(hypo, self.depth_interals_ratio[s], self.inverse_depth, logits=plan(cor),
                                       want_logits=want_logits)
            if capture is not None:
                capture[name] = {"cor_feats": cor.permute(0, 4, 1, 2, 3), "logits": sel["logits"],
                                 "feats_cl": f}                                  # [N,B,h,w,C], view 0 = reference
            # (x1 at the last stage is the identity, exactly: src = dst, lambda = 0.  The three coarse-stage launches stay on the
            #  main stream: on a stream of their own -- nothing in the cascade reads them -- the captured forward got SLOWER,
            #  1.174 against 1.129 ms alone and 850 against 1 108 depth-maps/s with two in flight, same box, alternating runs
            #  (profiles/r05_conf_stream_ab.txt): every cross-stream edge of a hipGraph costs more than these 4 us kernels)
            #  With `merge_launches` the three are ONE launch after the last stage.
            if s == 3 or self.merge_launches:
                conf = sel["conf"]
            else:
                conf = ops.upsample_bilinear(sel["conf"], 2 ** (3 - s))
            st = {"depth": sel["depth"], "photometric_confidence": conf, "hypo_depth": hypo,
                  "attn_weight": sel["attn_weight"]}
            if self.inverse_depth:
                st["inverse_min_depth"] = sel["inverse_min_depth"]
                st["inverse_max_depth"] = sel["inverse_max_depth"]
            if self.mono:
                st["mono_feat"] = ref_cl.permute(0, 3, 1, 2)                     # [B,C,h,w] view, channels-last memory
            prev = st
            outputs[name] = st
            outputs.update(st)
        if self.merge_launches:
            coarse = [outputs["stage%d" % (k + 1)] for k in range(min(3, self.num_stage))]
            ups = ops.upsample_bilinear_multi([st["photometric_confidence"] for st in coarse], H, W)
            for st, u in zip(coarse, ups):
                st["photometric_confidence"] = u
            if self.num_stage < 4:
                outputs["photometric_confidence"] = ups[-1]
        return outputs

    # ------------------------------------------------------------------ train: autograd
    def _stage_outputs(self, s, logits, hypo, dev):
        """Depth selection of one stage in training / differentiable form (models/mvs4net_utils.py:1068-1092)."""
        attn = F.softmax(logits, dim=1)
        idx = attn.max(1, keepdim=True)[1]
        depth = torch.gather(hypo, 1, idx).squeeze(1)
        if self.training:
            conf = torch.zeros((), dtype=torch.float32, device=dev)      # (torch.tensor(0.0, device=...) would sync)
        else:
            with torch.no_grad():
                conf = ops.upsample_bilinear(attn.max(1)[0].contiguous(), 2 ** (3 - s))
        st = {"depth": depth, "photometric_confidence": conf, "hypo_depth": hypo, "attn_weight": attn}
        if self.inverse_depth:
            itv = 1.0 / hypo[:, 2] - 1.0 / hypo[:, 1]
            st["inverse_min_depth"] = 1 / depth + self.depth_interals_ratio[s] * itv
            st["inverse_max_depth"] = 1 / depth - self.depth_interals_ratio[s] * itv
        return st

    @staticmethod
    def _no_stream_warning():
        if not MVS4net._stream_warning_off:
            # (DistributedDataParallel creates its AccumulateGrad nodes on the stream of its construction; gradients of the
            #  side streams' nodes reach them from those streams -- intended here, the engine synchronises the two)
            off = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
            if off is not None:
                off(False)
            MVS4net._stream_warning_off = True

    def _forward_train(self, imgs, proj_matrices, depth_values, teacher=None):
        """Differentiable forward with every convolution pass (forward, input and weight gradients) and the
        fused warp/correlation/aggregation on the gfx950 kernels.  The FPN runs once over all views (view-major
        batch) with BatchNorm on batch statistics per view, i.e. the numbers of the reference's per-view
        ``self.feature(img)`` calls (MVS4Net.py:65-68) in one pass.  ``teacher`` (tests): stage name -> hypotheses
        [B,D,h,w] to use instead of the ones scheduled from the previous stage's winners."""
        dev = imgs[0].device
        depth_values = depth_values.to(dev, torch.float32)
        depth_interval = (depth_values[:, -1] - depth_values[:, 0]) / depth_values.size(1)
        nv, B = len(imgs), imgs[0].shape[0]
        # all views through the FPN at once, view-major; BatchNorm statistics stay per view (groups = views)
        if any(img.requires_grad for img in imgs):
            x = torch.cat([img.permute(0, 2, 3, 1).unsqueeze(1) for img in imgs], 0)
        else:
            x = ops.pack_images([img.to(dev, torch.float32) for img in imgs])        # [N*B,1,H,W,4] RGB0, one launch
        # train_fpn_tail_stream: the FPN's two fine levels (needed from stage 3 on) on a stream of their own, beside stages 1-2
        tail = None
        if self.train_fpn_tail_stream and x.is_cuda and isinstance(self.feature, FPN4) and self.num_stage >= 3:
            tail = self._side_streams.get(("fpn_tail", dev))
            if tail is None:
                tail = self._side_streams[("fpn_tail", dev)] = torch.cuda.Stream(device=dev)
            self._no_stream_warning()
        if isinstance(self.feature, FPN4):
            self.feature.tail_stream = tail
        pyramid = self.feature.forward_cl(x, groups=nv)
        tail = getattr(self.feature, "tail_pending", None)
        if isinstance(self.feature, FPN4):
            self.feature.tail_stream = self.feature.tail_pending = None      # (direct callers of the module get one stream)
        outputs = {}
        prev = None
        ref_feats = []
        for s in range(self.num_stage):
            name = "stage%d" % (s + 1)
            if tail is not None and s >= 2:
                torch.cuda.current_stream(dev).wait_stream(tail)     # the fine levels are read from here on
                tail = None
            pyr = pyramid[name]                                      # [N*B,1,h,w,C], view-major
            _, _, h, w, C = pyr.shape
            G = self.group_cor_dim[s] if self.group_cor else C
            # train_side_stages: the stage's forward on a side stream (joined before anything else reads its outputs, so the
            # forward's order is unchanged) -- autograd then runs the stage's BACKWARD on that stream too, beside the other
            # stages' (the next stage detaches what it takes from this one: the stages' backward passes are independent)
            side = cur = None
            if s in self.train_side_stages and pyr.is_cuda:
                self._no_stream_warning()
                cur = torch.cuda.current_stream(dev)
                key = ("train", dev, s if self.train_side_separate else 0)
                side = self._side_streams.get(key)
                if side is None:
                    side = self._side_streams[key] = torch.cuda.Stream(device=dev)
                side.wait_stream(cur)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                with torch.no_grad():
                    rt = ops.relative_projection(proj_matrices[name].to(dev, torch.float32))
                    if teacher is not None and name in teacher:
                        hypo = teacher[name].detach().contiguous()
                    else:
                        hypo = self._hypotheses(s, depth_values, depth_interval, prev, h, w)
                cor, ref_maps = _WarpAggPyr.apply(pyr, B, rt, hypo, G, self.group_cor, self.attn_fuse_d, float(self.attn_temp))
                reg = self.reg[s]
                if isinstance(reg, reg2d) and self.training and self.stage_splits[s] >= (3 if self.inverse_depth else 1):
                    # prob head + softmax + argmax + gather + inverse bounds: one kernel forward, one backward
                    res = _SelectDepthCL.apply(reg.forward_cl(cor, return_features=True), reg.prob.weight, reg.prob.bias, hypo,
                                               float(self.depth_interals_ratio[s]), self.inverse_depth)
                    st = {"depth": res[1], "photometric_confidence": torch.zeros((), dtype=torch.float32, device=dev),
                          "hypo_depth": hypo, "attn_weight": res[0]}
                    if self.inverse_depth:
                        st["inverse_min_depth"], st["inverse_max_depth"] = res[2], res[3]
                else:
                    st = self._stage_outputs(s, reg.forward_cl(cor), hypo, dev)
            if side is not None:
                cur.wait_stream(side)
            if self.mono:
                st["mono_feat"] = ref_maps.reshape(B, h, w, C).permute(0, 3, 1, 2)     # [B,C,h,w] view
                ref_feats.append(ref_maps)
            prev = st
            outputs[name] = st
            outputs.update(st)
        if self.mono and self.training:
            outputs = self.mono_depth_decoder.forward_cl(outputs, ref_feats, depth_values[:, 0], depth_values[:, 1])
        return outputs

    def forward_eager(self, imgs, proj_matrices, depth_values):
        """The eval forward as plain launches on the current stream (what ``graph.GraphedForward`` captures and what
        ``forward`` runs on the first call of a shape)."""
        self._check_inputs(imgs, proj_matrices, depth_values)
        return self._forward_eval(imgs, proj_matrices, depth_values)

    def forward(self, imgs, proj_matrices, depth_values, filename=None):
        if self.training and max(self.stage_splits) > self.MAX_HYPOTHESES_TRAIN:
            raise NotImplementedError("training with stage_splits %r: the backward kernels (warp / aggregation, stage "
                                      "selection, Sinkhorn) hold at most %d hypotheses per pixel; evaluation takes up to %d"
                                      % (self.stage_splits, self.MAX_HYPOTHESES_TRAIN, self.MAX_HYPOTHESES))
        if self.training and min(self.stage_splits) < 3:
            # (MVS4net_loss reads hypotheses 1 and 2 of every stage for its range term, models/MVS4Net.py:139-144, in both
            #  depth modes: say so here, not from inside the loss after a whole forward)
            raise NotImplementedError("training with stage_splits %r: the loss (mvster_amd.loss.stage_losses, as "
                                      "MVS4net_loss in the reference) needs at least 3 hypotheses per stage; evaluation "
                                      "with inverse_depth=False takes 2" % (self.stage_splits,))
        self._check_inputs(imgs, proj_matrices, depth_values)
        if self.training:
            return self._forward_train(imgs, proj_matrices, depth_values)
        # nn.DataParallel over several devices hands every call a throw-away replica with freshly broadcast parameters:
        # nothing to cache there.  Inside somebody else's capture (GraphedForward) the launches are what gets recorded.
        if (not self.graph_cache or getattr(self, "_is_replica", False) or torch.cuda.is_current_stream_capturing()):
            return self._forward_eval(imgs, proj_matrices, depth_values)
        return self._fwd_cache(self, imgs, proj_matrices, depth_values)
