"""Differentiable channels-last building blocks of the training path, on the gfx950 kernels.

Training mode cannot fold BatchNorm (it runs on batch statistics), so the training forward is
conv -> BatchNorm(batch stats) -> ReLU layer by layer.  ``conv_cl`` is the convolution with all three
passes native:
  forward           mvster_conv_mfma / conv_small / deconv_small   (the inference kernels, scale 1, shift = bias)
  input gradient    the same kernels run as the adjoint convolution: flipped + transposed weights for a
                    stride-1 layer, the transposed (parity-class) form for a stride-2 layer, an ordinary
                    strided conv for a ConvTranspose layer
  weight gradient   mvster_conv_wgrad (voxels as the MFMA K dimension)
This replaces autograd through MIOpen, whose solvers take 30-50 ms per layer for these narrow,
spatially huge convolutions (profiles/r01_n_train_*).  BatchNorm + ReLU run as fused streaming kernels
(``batch_norm_cl``, statistics per view group); skip additions and upsampling are torch ops on the
channels-last tensors.

Reference: the layers of models/mvs4net_utils.py:116-123 (ConvBnReLU3D), :224-251 (Conv2d), :419-502 (FPN4),
:833-868 (mono_depth_decoder), :870-965 (reg2d / reg3d) under torch.autograd.
"""
import torch
import torch.nn.functional as F

from . import ops
from .conv_plan import ConvLayer

_ALLOWED_CIN = (4, 8, 16, 32, 64)


def _cin_for(c):
    for a in _ALLOWED_CIN:
        if c <= a:
            return a
    raise RuntimeError("conv_cl: more than 64 input channels (%d)" % c)


def _pad_last(x, c):
    return x if x.shape[-1] == c else F.pad(x, (0, c - x.shape[-1]))


def _triple(v, lead):
    v = tuple(v) if isinstance(v, (tuple, list)) else (v,) * 3
    return v if len(v) == 3 else (lead,) + v


class _EpochCells:
    """tensor -> epoch cell (a one-element list), held by identity WITHOUT keeping the tensor alive: entries die with
    their tensor (a weakref callback removes them), and a lookup checks that the entry still belongs to the tensor
    asked about -- a later tensor that reuses the id of a dead one never inherits its cell."""

    def __init__(self):
        self._d = {}

    def __bool__(self):
        return bool(self._d)

    def __len__(self):
        return len(self._d)

    def set(self, tensor, cell):
        import weakref
        key = id(tensor)
        d = self._d

        def gone(_ref, key=key):
            hit = d.get(key)
            if hit is not None and hit[0] is _ref:
                del d[key]
        d[key] = (weakref.ref(tensor, gone), cell)

    def get(self, tensor):
        hit = self._d.get(id(tensor))
        if hit is None or hit[0]() is not tensor:
            return None
        return hit[1]

    def drop(self, cell):
        """Forget every tensor registered under ``cell`` (a GraphedTrainStep going away)."""
        for k in [k for k, v in self._d.items() if v[1] is cell]:
            del self._d[k]

    def clear(self):
        self._d.clear()


class _LayerCache:
    """ConvLayer objects per (parameter, role): built once, re-packed when the parameter changes, so a training step
    costs a few small device ops per layer instead of a plan build.

    "Changed" = another ``_version`` (in-place updates through the tensor: torch.optim steps, ``p.mul_()``) or another
    storage address (``p.data = t``, ``vector_to_parameters``).  In-place writes through ``p.data`` / a detached alias
    (``p.data.add_()``, some third-party optimizers, EMA helpers) move neither: they cannot be seen from here, so either
    call ``train_ops.CACHE.clear()`` / ``model.invalidate_plans()`` after such an update or set
    ``train_ops.CACHE.always_repack = True`` (one small kernel per layer and pass; already what a captured
    ``GraphedTrainStep`` replays every step).  An optimizer update that runs inside a hipGraph replay is such a write too
    (the replay bumps no ``_version``): a ``GraphedTrainStep`` registers ONE epoch cell (a one-element list) under the id
    of every parameter and buffer of its model in ``CACHE.cells`` and bumps it after every replay; the cell's value is part
    of the stamp of those tensors here and in ``MVS4net._state_stamp``, so the next eager forward of THAT model --
    training or eval -- re-packs, and other models in the process are left alone."""

    def __init__(self):
        self._d = {}
        self.cells = _EpochCells()     # parameter / buffer -> [epoch]; see above
        self.always_repack = False
        # bumped whenever a training-mode BatchNorm writes running statistics through raw pointers (ops.bn_batch_stats moves
        # no ``_version``): part of ``MVS4net._state_stamp``, so an eval graph folded from the old statistics is not replayed
        self.stat_writes = 0
        self._batch = None             # see build_batch()
        self._batch_active = False

    def get(self, weight, role, build, refresh, owner=None, rargs=None, bias=None):
        """``build()`` makes the layer; ``refresh(layer)`` re-packs it from the updated parameter.  ``owner``: the
        parameter a derived weight (a product of parameters, a new tensor every step) is cached under; such layers are
        re-packed on every call.  ``rargs`` = the (swap, flip) that ``refresh`` hands to ``ConvLayer.repack_on_device``
        (None: the refresh is not of that form and the layer stays out of the batched refresh).  ``bias``: the layer's bias
        parameter, copied into the epilogue's shift vector at every call (or by the batched refresh)."""
        derived = owner is not None
        if derived:
            weight = owner
        key = (id(weight), role)
        hit = self._d.get(key)
        cell = self.cells.get(weight) if self.cells else None
        stamp = (weight._version, weight.data_ptr(), 0 if cell is None else cell[0])
        if hit is not None and hit[0] is weight and hit[2].wpk.device == weight.device:
            if self._batch_active and key in self._batch["keys"]:
                return hit[2]                  # refreshed (bias included) by run_batch() at the top of this step
            if derived or hit[1] != stamp or self.always_repack or torch.cuda.is_current_stream_capturing():
                refresh(hit[2])
                self._d[key] = (weight, stamp, hit[2], hit[3], bias)
            if bias is not None:
                hit[2].shift[:hit[2].cout] = bias.detach()
            return hit[2]
        layer = build()
        if bias is not None:
            layer.shift[:layer.cout] = bias.detach()
        self._d[key] = (weight, stamp, layer, None if derived else rargs, bias)
        return layer

    def clear(self):
        self._d.clear()
        self._batch, self._batch_active = None, False

    # ---- batched refresh (captured training steps) ------------------------------------------------------------------------
    # Inside a hipGraph every cached layer is re-packed at every use (the parameters have moved, and the capture must
    # record the refresh): ~130 launches of ~3 us per step.  Every packed / permuted form is a fixed permutation (+ zero
    # padding) of its parameter, so it is recorded ONCE -- by running the layer's own refresh on a tensor that holds its
    # own indices -- and a step replays all of them in one launch of mvster_gather_batch (the Winograd-transformed
    # weights, which are arithmetic, keep their own small kernel).
    def build_batch(self, params=None):
        """Record the permutations of every cached layer that is refreshed through ``ConvLayer.repack_on_device`` from a
        contiguous parameter (of ``params``, if given) and return them as one object for ``run_batch`` -- owned by the
        caller: a captured graph replays a launch that reads its tables.  Call after a step has populated the cache
        (GraphedTrainStep does, after its warm-up).  None if there is nothing to batch."""
        import numpy as np
        from . import _lib
        only = None if params is None else {id(p) for p in params}
        recs, keep, keys, wino = [], [], set(), []
        for key, (weight, _stamp, layer, rargs, bias) in self._d.items():
            if rargs is None or not weight.is_cuda or not weight.is_contiguous() or weight.dtype != torch.float32:
                continue
            if only is not None and id(weight) not in only:
                continue
            if weight.numel() >= (1 << 24):
                continue                       # (indices travel through fp32)
            swap, flip = rargs
            bufs = [b for b in (layer.wpk, layer.w_small, layer.w_deconv) if b is not None]
            if any(b.numel() % 4 or not b.is_contiguous() for b in bufs):
                continue
            if bias is not None and not (bias.is_contiguous() and bias.dtype == torch.float32 and layer.shift.numel() % 4 == 0):
                continue                   # (this layer keeps its per-call refresh)
            iota = torch.arange(1, weight.numel() + 1, device=weight.device, dtype=torch.float32).view(weight.shape)
            had_wino, layer.wpk_wino = layer.wpk_wino, None          # (arithmetic, not a permutation: not on the iota pass)
            layer.repack_on_device(iota, swap=swap, flip=flip)
            idx = [b.round().to(torch.int32).clone() for b in bufs]
            layer.wpk_wino = had_wino
            layer.repack_on_device(weight, swap=swap, flip=flip)     # back to the real weights
            for b, ix in zip(bufs, idx):
                recs.append((weight.data_ptr(), b.data_ptr(), ix.data_ptr(), b.numel()))
                keep.append(ix)
            if bias is not None:
                ix = torch.zeros(layer.shift.numel(), device=weight.device, dtype=torch.int32)
                ix[:layer.cout] = torch.arange(1, layer.cout + 1, device=weight.device, dtype=torch.int32)
                recs.append((bias.data_ptr(), layer.shift.data_ptr(), ix.data_ptr(), ix.numel()))
                keep.append(ix)
            if layer.wpk_wino is not None:
                wino.append((layer, weight, swap, flip))
            keys.add(key)
        if not recs:
            return None
        dev = keep[0].device
        table = np.zeros(len(recs), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("idx", "<u8"), ("n", "<i4"), ("fb", "<i4")]))
        fb = 0
        for i, (s_, d_, x_, n_) in enumerate(recs):
            table[i] = (s_, d_, x_, n_, fb)
            fb += (n_ + 1023) // 1024
        descs = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
        # the Winograd-transformed forms (arithmetic, not permutations): one table for mvster_pack_wino_batch
        wdescs, wblocks = None, 0
        if wino:
            wt = np.zeros(len(wino), dtype=np.dtype([("w", "<u8"), ("wpk", "<u8"), ("s_n", "<i8"), ("s_c", "<i8"), ("s_z", "<i8"),
                                                     ("s_y", "<i8"), ("s_x", "<i8"), ("cout", "<i4"), ("cin_raw", "<i4"),
                                                     ("cin", "<i4"), ("kd", "<i4"), ("flip", "<i4"), ("ntile", "<i4"),
                                                     ("fb", "<i4"), ("pad", "<i4")]))
            assert wt.dtype.itemsize == 88
            for i, (layer, weight, swap, flip) in enumerate(wino):
                w = weight if weight.dim() == 5 else weight.unsqueeze(2)
                st = w.stride()
                s_n, s_c = (st[1], st[0]) if swap else (st[0], st[1])
                ntile = (layer.cout + 15) // 16
                wt[i] = (w.data_ptr(), layer.wpk_wino.data_ptr(), s_n, s_c, st[2], st[3], st[4], layer.cout, layer._cin_raw,
                         layer.cin, layer.kernel[0], int(flip), ntile, wblocks, 0)
                wblocks += (layer.kernel[0] * 16 * layer.cin * ntile * 16 + 255) // 256
            wdescs = torch.from_numpy(wt.view(np.uint8).copy()).to(dev)
        return {"keys": keys, "descs": descs, "n": len(recs), "blocks": fb, "idx": keep, "wino": wino, "lib": _lib,
                "wino_descs": wdescs, "wino_blocks": wblocks}

    def run_batch(self, b):
        """Refresh every form recorded in ``b`` from its parameter (one launch + the Winograd transforms) and let ``get``
        skip the per-layer refresh of those layers until ``end_batch``."""
        if b is None:
            return
        self._batch = b
        rc = b["lib"].load().mvster_gather_batch(b["descs"].data_ptr(), b["n"], b["blocks"], ops._stream())
        b["lib"].check(rc, "gather_batch")
        if b["wino_descs"] is not None:
            rc = b["lib"].load().mvster_pack_wino_batch(b["wino_descs"].data_ptr(), len(b["wino"]), b["wino_blocks"], ops._stream())
            b["lib"].check(rc, "pack_wino_batch")
        self._batch_active = True

    def end_batch(self):
        self._batch_active = False
        self._batch = None


CACHE = _LayerCache()


class _ConvCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, transposed, skip=None, skip_upsample=False, owner=None, tap=False):
        w5 = weight if weight.dim() == 5 else weight.unsqueeze(2)
        own, tag = owner if owner is not None else (None, "")
        cin = w5.shape[0] if transposed else w5.shape[1]
        cin_p = _cin_for(cin)
        # (an input that already carries the kernels' channel count is taken as it is -- the RGB0 batch of
        #  ops.pack_images for the 3-channel first layer; its extra channels meet zero weights)
        if x.shape[-1] not in (cin, cin_p):
            raise RuntimeError("conv_cl: input has %d channels, weight expects %d" % (x.shape[-1], cin))
        xp = _pad_last(x, cin_p).contiguous()
        layer = CACHE.get(weight, "fwd" + tag, lambda: ConvLayer(w5, transposed, stride, padding, cin_pad=cin_p),
                          lambda L: L.repack_on_device(weight), owner=own, rargs=(False, False), bias=bias)
        if skip is not None:
            # y = conv(x) + skip, or + the bilinear x2 (align_corners) of a half-resolution skip, in the kernel's epilogue:
            # the FPN's top-down sums (mvs4net_utils.py:488-496) without materialising the up-sampled 64-channel map
            from .conv_plan import SKIP_ADD, SKIP_UPSAMPLE_ADD
            y = layer(xp, skip=skip.contiguous(), skip_mode=SKIP_UPSAMPLE_ADD if skip_upsample else SKIP_ADD)
        else:
            y = layer(xp)
        ctx.has_skip = (skip is not None, skip_upsample)
        ctx.save_for_backward(xp, weight, bias)
        ctx.cfg = (stride, padding, transposed, cin, x.shape[-1])
        ctx.owner = (own, tag)
        ctx.tap = tap
        if tap:
            # second output: x itself, for the OTHER consumers of x (a U-Net skip connection, an FPN lateral).  Their gradient
            # comes back here and is added in the input-gradient kernel's epilogue -- instead of the autograd engine's
            # accumulate, a full-size read-read-write launch per fan-out (18 per step, 0.39 ms at 512x640x5, B = 2)
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, gy, gtap=None):
        if gy is None:                     # (only the tap was used)
            return (gtap,) + (None,) * 9
        xp, weight, bias = ctx.saved_tensors
        stride, padding, transposed, cin, x_channels = ctx.cfg
        own, tag = ctx.owner
        w5 = weight if weight.dim() == 5 else weight.unsqueeze(2)
        kernel = tuple(w5.shape[2:])
        gy = gy.contiguous()
        gx = gw = gb = None
        co = gy.shape[-1]
        co_p = _cin_for(co)
        # the output gradient padded to a channel count the kernels take (1 -> 4 for the depth heads); also what the
        # weight gradient reads: with a multiple of 4 channels it runs on the LDS-staged kernel
        gyp = _pad_last(gy, co_p).contiguous() if co != co_p else gy
        if ctx.needs_input_grad[0]:
            if transposed:
                # y = convT(x; W[cin,cout]) : dx = conv(gy; W read as [out=cin, in=cout], same stride / padding)
                layer = CACHE.get(weight, "dgrad" + tag, lambda: ConvLayer(w5, False, stride, padding, cin_pad=co_p),
                                  lambda L: L.repack_on_device(weight), owner=own, rargs=(False, False))
            elif stride == (1, 1, 1):
                def flipped():
                    return w5.detach().flip(2, 3, 4).transpose(0, 1).contiguous()
                pad = tuple(k - 1 - p for k, p in zip(kernel, padding))
                layer = CACHE.get(weight, "dgrad" + tag, lambda: ConvLayer(flipped(), False, stride, pad, cin_pad=co_p),
                                  lambda L: L.repack_on_device(weight, swap=True, flip=True), owner=own, rargs=(True, True))
            else:
                # stride 2: the adjoint is the transposed conv with the same weights (parity classes)
                layer = CACHE.get(weight, "dgrad" + tag, lambda: ConvLayer(w5, True, stride, padding, cin_pad=co_p),
                                  lambda L: L.repack_on_device(weight), owner=own, rargs=(False, False))
            fused_tap = gtap is not None and layer.cout == x_channels and tuple(gtap.shape[:4]) == tuple(xp.shape[:4])
            if fused_tap:
                from .conv_plan import SKIP_ADD
                gx = layer(gyp, skip=gtap.contiguous(), skip_mode=SKIP_ADD)
            else:
                gx = layer(gyp)
            if tuple(gx.shape[:4]) != tuple(xp.shape[:4]):
                raise RuntimeError("conv_cl: input gradient of a strided layer needs even input sizes (%s -> %s)"
                                   % (tuple(xp.shape), tuple(gx.shape)))
            gx = gx[..., :x_channels]
            if gtap is not None and not fused_tap:
                gx = gx + gtap
        elif gtap is not None:
            gx = gtap
        if ctx.needs_input_grad[1]:
            # (a derived weight -- the finest FPN level's composed 3x3 -- hands its gradient on inside this backward pass)
            leaf = own is None and weight.is_leaf
            if transposed:
                wargs = (gyp, xp, kernel, stride, padding)                                   # -> [cin, cout, k]
                wkw = dict(co_keep=cin, ci_keep=co)
            elif stride == (1, 1, 1) and co <= 8 < xp.shape[-1] and kernel != (1, 1, 1):
                # narrow OUTPUT side: the mirrored sum  dW[co][ci][t] = sum_q x[q][ci] * gy[q + p - t][co]  has gy as the
                # shifted tensor, so the tap-packed kernel applies with the roles swapped (taps and padding mirrored)
                mirror = tuple(k - 1 - p for k, p in zip(kernel, padding))
                wargs = (gyp, xp, kernel, stride, mirror)
                wkw = dict(co_keep=cin, ci_keep=co, mirrored=True)
            else:
                wargs = (xp, gyp, kernel, stride, padding)                                   # -> [cout, cin, k]
                wkw = dict(co_keep=co, ci_keep=cin)
            if leaf and _WGRAD_JOBS is not None:
                # deferred_wgrad_finish: nothing reads this gradient before the backward pass is over, so the kernel itself
                # waits for the end of the pass and runs there beside the other layers' (see the context manager)
                dw = torch.empty(ops.conv_wgrad_shape(wargs[0], wargs[1], kernel, **wkw), device=gy.device, dtype=torch.float32)
                _WGRAD_JOBS.append((wargs, wkw, dw, torch.cuda.current_stream(gy.device) if gy.is_cuda else None))
                if _WGRAD_EAGER is not None:
                    _WGRAD_EAGER.launch(wargs, wkw, dw)
                gw = dw.view(dw.shape)           # (an alias: AccumulateGrad adopts a tensor only if nobody else holds it)
            else:
                gw = ops.conv_wgrad(*wargs, **wkw)
            if weight.dim() == 4:
                gw = gw.squeeze(2)
        if bias is not None and ctx.needs_input_grad[2]:
            # (the padded gradient where the head has a single output channel: the kernel sums 4 / 8 / ... / 64 columns)
            gb = ops.col_sum(gyp)[:co] if co_p in (4, 8, 16, 32, 64) and gy.numel() else gy.sum((0, 1, 2, 3))
        gskip = None
        if ctx.has_skip[0]:
            gskip = gy
            if ctx.has_skip[1]:       # adjoint of the bilinear x2 (gather form, no atomics)
                B_, D_, H_, W_, C_ = gy.shape
                gskip = ops.upsample2x_cl(gy.reshape(B_ * D_, H_, W_, C_), backward=True).reshape(B_, D_, H_ // 2, W_ // 2, C_)
        return gx, gw, gb, None, None, None, gskip, None, None, None


def conv_cl(x, weight, bias=None, stride=1, padding=0, transposed=False, skip=None, skip_upsample=False, owner=None, tap=False):
    """x [B,D,H,W,Cin] channels-last -> [B,Do,Ho,Wo,Cout]; weight in nn.Conv3d / nn.Conv2d / nn.ConvTranspose3d
    layout (a 4-D weight is a depth-1 convolution).  ``skip`` is added in the kernel's epilogue: a tensor of the output's
    shape, or with ``skip_upsample`` a [B,1,Ho/2,Wo/2,Cout] map that is bilinearly up-sampled x2 (align_corners) on the fly.
    ``tap``: -> (y, x') with x' an alias of x for x's other consumers: what flows back into x' is added to the input
    gradient inside the input-gradient kernel (use x' INSTEAD of x downstream, or the fan-out costs an add launch again)."""
    lead_s, lead_p = 1, 0
    return _ConvCL.apply(x, weight, bias, _triple(stride, lead_s), _triple(padding, lead_p), transposed, skip, skip_upsample,
                         owner, tap)


def tap_bias_grad(gP):
    """Gradient of the gather-sum's per-tap bias terms vb [9, co] given gP [NB,1,H,W,co]: d/d vb[tap] = the sum of gP over
    the pixels whose tap lies inside the image -- everything, minus the first / last row and / or column."""
    g = gP[:, 0]
    co = g.shape[-1]
    rows = g.sum(2)                                                              # over x: [NB,H,co]
    xs = torch.stack([rows - g[:, :, 0], rows, rows - g[:, :, -1]])             # kx = 0, 1, 2
    tot = xs.sum(2)                                                              # over y: [3,NB,co]
    return torch.stack([tot - xs[:, :, 0], tot, tot - xs[:, :, -1]]).sum(2).reshape(9, co)   # [ky,kx] -> [9,co]


class _FpnGather(torch.autograd.Function):
    """P = gather-sum of the nine shifted bilinear x2 up-samplings of G = conv1x1(f; wg) (+ the bias terms vb): the
    top-down half of a re-associated FPN level (see ``fpn_fine_level``).  All passes on the gfx950 kernels; the 72-channel
    gradient of G is written with an 80-channel pitch so that the input-gradient conv and the weight gradient read it
    as it is."""

    PITCH = {8: 80}          # 9*co rounded up to the input-gradient conv's channel granularity

    @staticmethod
    def forward(ctx, f, wg, vb, H, W, owner):
        w5 = wg.unsqueeze(2)
        layer = CACHE.get(wg, "fpn_g", lambda: ConvLayer(w5, False, (1, 1, 1), (0, 0, 0)), lambda L: L.repack_on_device(wg),
                          owner=owner)
        f = f.contiguous()
        P = ops.fpn_tail_gather(layer(f), vb.detach().contiguous(), H, W)
        ctx.save_for_backward(f, wg)
        ctx.owner = owner
        return P

    @staticmethod
    def backward(ctx, gP):
        f, wg = ctx.saved_tensors
        gP = gP.contiguous()
        co = gP.shape[-1]
        if co not in _FpnGather.PITCH:
            raise NotImplementedError("fpn_fine_level: %d output channels (the path's finest level has 8)" % co)
        pitch = _FpnGather.PITCH[co]
        gG = ops.fpn_tail_gather_bwd(gP, pitch=pitch)                       # [NB,1,h,w,pitch]
        gf = gwg = gvb = None
        if ctx.needs_input_grad[0]:
            def transposed_weight():                                        # [64 out, 9co in] -> padded to `pitch` inputs
                return wg.detach()[:, :, 0, 0].t().reshape(wg.shape[1], wg.shape[0], 1, 1, 1).contiguous()
            layer = CACHE.get(wg, "fpn_g_dgrad", lambda: ConvLayer(transposed_weight(), False, (1, 1, 1), (0, 0, 0), cin_pad=pitch),
                              lambda L: L.repack_on_device(wg, swap=True), owner=ctx.owner)
            gf = layer(gG)
        if ctx.needs_input_grad[1]:
            gwg = ops.conv_wgrad(f, gG, (1, 1, 1), (1, 1, 1), (0, 0, 0), co_keep=wg.shape[0]).reshape(wg.shape)
        if ctx.needs_input_grad[2]:
            gvb = tap_bias_grad(gP)
        return gf, gwg, gvb, None, None, None


class _FineWeights(torch.autograd.Function):
    """(wg, wc, vb) of ``fpn_fine_level`` from out.weight, inner.weight, inner.bias: one single-workgroup launch each way
    (as einsum / permute / reshape expressions: ~20 launches per step, three of them rocBLAS calls)."""

    @staticmethod
    def forward(ctx, wo, wi, bi):
        wo_c, wi_c, bi_c = wo.detach().contiguous(), wi.detach().contiguous(), bi.detach().contiguous()
        ctx.save_for_backward(wo_c, wi_c, bi_c)
        ctx.set_materialize_grads(False)
        return ops.fine_weights_fwd(wo_c, wi_c, bi_c)

    @staticmethod
    def backward(ctx, g_wg, g_wc, g_vb):
        wo, wi, bi = ctx.saved_tensors

        def c(g):
            return None if g is None else g.contiguous()
        if g_wg is None and g_wc is None and g_vb is None:
            return None, None, None
        return ops.fine_weights_bwd(wo, wi, bi, c(g_wg), c(g_wc), c(g_vb))


def fpn_fine_level(c_low, f_coarse, inner, out):
    """out(F.interpolate(f_coarse, x2 bilinear, align_corners) + inner(c_low)) of the FPN's top-down path
    (models/mvs4net_utils.py:488-489) WITHOUT forming the 64-channel map at c_low's resolution (839 MB for ten 512x640
    views, plus its gradient): ``out`` is linear and interpolation acts per channel, so
        out(f)[p] = sum_tap up(W_out[tap] f_coarse)[p + tap] + sum_tap W_out[tap] (W_in c_low[p + tap] + b_in)
    = the gather-sum of a 1x1 conv of f_coarse (_FpnGather) + a 3x3 conv of c_low with the composed weights; the composed
    weights are differentiable functions of the parameters, so autograd carries the gradients back to ``out.weight``,
    ``inner.weight`` and ``inner.bias``.  Same re-association as the inference plan (conv_plan.FpnPlan)."""
    # wg [9co, 64, 1, 1] (row = tap*co + co_idx) = out.weight permuted; wc [co, cin, 3, 3] = out.weight x inner.weight;
    # vb [9, co] = out.weight x inner.bias
    wg, wc, vb = _FineWeights.apply(out.weight, inner.weight, inner.bias)
    H, W = c_low.shape[2], c_low.shape[3]
    P = _FpnGather.apply(f_coarse, wg, vb, H, W, out.weight)
    return conv_cl(c_low, wc, out.bias, out.stride, out.padding, skip=P, owner=(out.weight, "_fine_c"))


class _BnReluCL(torch.autograd.Function):
    """relu(BatchNorm(x)) given the statistics pack [5, groups, C] = (mean, var, rstd, scale, shift): fused apply kernel
    forward, two fused kernels backward.  ``frozen``: the pack holds the running statistics (a BatchNorm in eval mode
    inside a training graph), which do not depend on x."""

    @staticmethod
    def forward(ctx, x, weight, bias, pack, relu, groups, frozen, skip=None):
        ctx.save_for_backward(x, pack)
        ctx.cfg = (relu, groups, frozen, skip is not None)
        return ops.bn_relu_fwd(x, pack[3], pack[4], relu, groups, skip=None if skip is None else skip.contiguous())

    @staticmethod
    def backward(ctx, gy):
        x, pack = ctx.saved_tensors
        relu, groups, frozen, has_skip = ctx.cfg
        gy = gy.contiguous()
        C = x.shape[-1]
        if not frozen and ops.bn_fused_ok(x.numel() // C // groups, C, groups, True):
            dx, dbeta, dgamma = ops.bn_bwd_fused(x, gy, pack, relu, groups)          # (small tensor: one launch)
        elif not frozen and ops.BN_TAILLESS:
            dx, dbeta, dgamma = ops.bn_train_bwd(x, gy, pack, relu, groups)          # (slot sums in the apply kernel's prologue)
        else:
            dx, dbeta, dgamma = ops.bn_relu_bwd(x, gy, pack[3], pack[4], pack[0], pack[2], relu, groups, frozen)
        gskip = gy if has_skip else None          # the skip connection's gradient is the output gradient itself
        return dx, dgamma, dbeta, None, None, None, None, gskip


class _BnFusedCL(torch.autograd.Function):
    """Training-mode relu(BatchNorm(x)) (+ skip) with the statistics finished INSIDE the forward op: ``mvster_bn_train_fwd``
    (slots, then the apply kernel sums them in its prologue) or, behind ``ops.BN_FUSED``, ``mvster_bn_fwd_fused`` (a small
    tensor in one launch); the backward is ``_BnReluCL``'s, which picks the matching form."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, nbt, eps, momentum, relu, groups, skip=None):
        C = x.shape[-1]
        fn = ops.bn_fwd_fused if ops.bn_fused_ok(x.numel() // C // groups, C, groups, False) else ops.bn_train_fwd
        y, pack = fn(x, weight.detach(), bias.detach(), running_mean, running_var, eps, momentum, relu, groups,
                     num_batches_tracked=nbt, skip=None if skip is None else skip.contiguous())
        ctx.save_for_backward(x, pack)
        ctx.cfg = (relu, groups, False, skip is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        dx, dgamma, dbeta, _, _, _, _, gskip = _BnReluCL.backward(ctx, gy)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, gskip


def batch_norm_cl(x, bn, relu=False, groups=1, skip=None):
    """nn.BatchNorm2d / 3d (+ optional ReLU) on a channels-last tensor, on the fused gfx950 kernels (mvster_bn_*):
    batch statistics + running-stat update in training (torch semantics: biased variance to normalise, unbiased in the
    running average), running statistics when the module is in eval mode.  ``groups`` > 1: x holds that many equal
    slices along dim 0 (the views of one sample batch) which are normalised separately and update the running
    statistics one after the other, exactly as ``groups`` separate calls of the module would (the reference runs its
    FPN once per view).  ``skip``: a tensor of x's shape added after the activation, in the same kernel (the U-Net's skip
    connections).  Gradients flow to x, gamma, beta (and skip).  Configurations the kernels do not cover raise."""
    C = x.shape[-1]
    if not x.is_cuda:
        raise RuntimeError("batch_norm_cl: expected a GPU tensor (the HIP path has no CPU fallback)")
    if C not in (4, 8, 16, 32, 64):
        raise NotImplementedError("batch_norm_cl: %d channels (the kernels cover 4/8/16/32/64, the path's widths)" % C)
    if not bn.affine:
        raise NotImplementedError("batch_norm_cl: non-affine BatchNorm (not used by the path)")
    x = x.contiguous()
    batch_stats = bn.training or not bn.track_running_stats
    if batch_stats:
        track = bn.track_running_stats
        if track and bn.momentum is None:
            raise NotImplementedError("batch_norm_cl: cumulative-average running statistics (momentum=None)")
        if track:
            CACHE.stat_writes += 1
        if ops.BN_TAILLESS or ops.bn_fused_ok(x.numel() // C // groups, C, groups, False):
            # statistics slots + apply with the finish in its prologue (or, behind ops.BN_FUSED, a small tensor in ONE launch)
            return _BnFusedCL.apply(x, bn.weight, bn.bias, bn.running_mean if track else None, bn.running_var if track else None,
                                    bn.num_batches_tracked if track else None, bn.eps, bn.momentum or 0.0, relu, groups, skip)
        with torch.no_grad():
            pack = ops.bn_batch_stats(x, bn.weight, bn.bias, bn.running_mean if track else None,
                                      bn.running_var if track else None, bn.eps, bn.momentum or 0.0, groups,
                                      num_batches_tracked=bn.num_batches_tracked if track else None)
        return _BnReluCL.apply(x, bn.weight, bn.bias, pack, relu, groups, False, skip)
    with torch.no_grad():            # [C]-sized parameter preparation, like weight packing
        rstd = torch.rsqrt(bn.running_var + bn.eps)
        scale = bn.weight * rstd
        pack = torch.stack([bn.running_mean, bn.running_var, rstd, scale, bn.bias - bn.running_mean * scale])
        pack = pack.unsqueeze(1).expand(5, groups, C).contiguous()
    return _BnReluCL.apply(x, bn.weight, bn.bias, pack, relu, groups, True, skip)


class _Upsample2xCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        return ops.upsample2x_cl(x.contiguous(), mode=mode)

    @staticmethod
    def backward(ctx, g):
        return ops.upsample2x_cl(g.contiguous(), backward=True, mode=ctx.mode), None


def upsample2x_cl(x, mode):
    """F.interpolate(scale_factor=2) of a [B,D,H,W,C] channels-last map (per depth slice): "bilinear"
    (align_corners=True, the FPN's top-down path) or "nearest" (the mono head), each a gfx950 kernel pair with a
    gather-form adjoint (no atomics)."""
    B, D, H, W, C = x.shape
    return _Upsample2xCL.apply(x.reshape(B * D, H, W, C), mode).reshape(B, D, 2 * H, 2 * W, C)


class _UpCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)          # (shape donors; both are alive as other nodes' saved tensors anyway)
        return ops.upcat(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        return ops.upcat(a, b, backward=g.contiguous())


def upcat_cl(a, b):
    """torch.cat([nearest x2 of a, b], -1) for channels-last maps a [NB,1,H/2,W/2,Ca], b [NB,1,H,W,Cb] -- the input of the
    monocular head's 3x3 convolutions (models/mvs4net_utils.py:854-857) -- one launch forward, one backward."""
    return _UpCat.apply(a, b)


class _MonoDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, dmin, dmax):
        dmin, dmax = dmin.detach().to(torch.float32).contiguous(), dmax.detach().to(torch.float32).contiguous()
        depth, sig = ops.mono_depth_fwd(z.contiguous(), dmin, dmax)
        ctx.save_for_backward(depth, sig, dmin, dmax)
        return depth

    @staticmethod
    def backward(ctx, g):
        depth, sig, dmin, dmax = ctx.saved_tensors
        return ops.mono_depth_bwd(g.contiguous(), depth, sig, dmin, dmax), None, None


def mono_depth_cl(z, d_min, d_max):
    """1 / (1/d_max + (1/d_min - 1/d_max) * sigmoid(z)) per sample (models/mvs4net_utils.py:858-866), z of any shape
    [B, ...]; one launch forward, one backward (as tensor expressions: ~13 forward and ~10 backward launches per stage)."""
    return _MonoDepth.apply(z, d_min, d_max)


_WGRAD_JOBS = None           # a list while deferred_wgrad_finish is active: (args, kwargs, dw) of the postponed kernels
_WGRAD_EAGER = None          # the active deferred_wgrad_finish(overlap=True): kernels go to its side stream where autograd reaches them
_WGRAD_STREAMS = {}
_WGRAD_CTX = None            # the active deferred_wgrad_finish (for wgrad_flush_point)


class _FlushPoint(torch.autograd.Function):
    """Identity; its backward tells the active ``deferred_wgrad_finish(early=True)`` to launch what it has collected."""

    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        if _WGRAD_CTX is not None and _WGRAD_CTX.early:
            _WGRAD_CTX.flush_now()
        return g


def wgrad_flush_point(t):
    """Mark a point of the backward pass at which ``deferred_wgrad_finish(early=True)`` launches the weight-gradient kernels
    postponed so far: FPN4 puts it on its finest output, which autograd reaches when every cascade stage's backward has been
    issued and before the FPN's begins -- the regularisation networks' weight gradients then run beside the FPN's backward
    chain instead of after it (one fork in the captured step, not one per layer)."""
    return _FlushPoint.apply(t) if t.requires_grad else t


class deferred_wgrad_finish:
    """Context manager around a backward pass whose weight gradients nobody reads before it ends (no hook-driven reducer
    such as DistributedDataParallel's; ``GraphedTrainStep`` wraps its backward in it).  The weight-gradient kernels of leaf
    parameters are not launched where autograd reaches them but on exit, round-robin over ``streams`` HIP streams forked
    from the current one and joined back; every stream finishes its layers' partial sums with ONE batched launch (64
    finishing launches of 5 us otherwise).  ``overlap=True`` launches each kernel on one side stream where autograd reaches
    its layer instead (beside the rest of the backward chain; the finishes stay batched at the end).  Measured on the
    config-4 step: after the chain on two streams is the fastest form (one until the cascade stages' backward
    moved to a side stream) -- the kernels are persistent grids sized for an empty chip, and every cross-stream edge of the
    captured graph costs ~10 us.  The activations and output
    gradients the kernels read are kept alive until the join.  Plain ``loss.backward()`` outside the context runs every
    layer on the spot."""

    def __init__(self, streams=1, overlap=False, policy="rr", early=False):
        self.nstreams = max(1, int(streams))
        self.overlap = bool(overlap)
        self.early = bool(early) and not self.overlap    # launch in two batches: at wgrad_flush_point and on exit
        self.policy = policy             # "rr": round-robin in autograd's order; "lpt": longest estimated job to the least loaded stream
        self._side = None
        self._pend = []

    def __enter__(self):
        global _WGRAD_JOBS, _WGRAD_EAGER
        if _WGRAD_JOBS is not None:
            raise RuntimeError("deferred_wgrad_finish does not nest")
        global _WGRAD_CTX
        _WGRAD_JOBS = []
        _WGRAD_EAGER = self if self.overlap else None
        _WGRAD_CTX = self
        self._side, self._pend = None, []
        self._launched, self._sides, self._pends = 0, None, None
        return self

    def _streams(self, dev):
        if self._sides is None:
            pool = _WGRAD_STREAMS.setdefault(dev, [])
            while len(pool) < self.nstreams:
                pool.append(torch.cuda.Stream(device=dev))
            self._sides = pool[:self.nstreams]
            self._pends = [[] for _ in self._sides]
        return self._sides

    def flush_now(self):
        """Launch the jobs collected since the last call on the side streams (forked from the streams the jobs were queued
        on); their finishes stay for the batched launch on exit."""
        jobs = _WGRAD_JOBS[self._launched:] if _WGRAD_JOBS else []
        if not jobs:
            return
        dev = jobs[0][2].device
        side = self._streams(dev)
        first = self._launched
        self._launched += len(jobs)
        sources = []
        for j in jobs:
            if j[3] is not None and all(j[3] != q for q in sources):
                sources.append(j[3])
        cur = torch.cuda.current_stream(dev)
        if all(cur != q for q in sources):
            sources.append(cur)
        for s_ in side:
            for q in sources:
                s_.wait_stream(q)                                        # fork: every input of these jobs is ready
        where = [(first + i) % len(side) for i in range(len(jobs))]
        if self.policy == "lpt" and len(side) > 1:
            def cost(job):
                x_, gy_, kern = job[0][0], job[0][1], job[0][2]
                flops = 2.0 * gy_.numel() * x_.shape[-1] * kern[0] * kern[1] * kern[2]
                return max(flops / 40e12, 4.0 * (x_.numel() + gy_.numel()) / 3e12) + 4e-6
            load = [0.0] * len(side)
            for i, job in sorted(enumerate(jobs), key=lambda t: -cost(t[1])):
                k = load.index(min(load))
                where[i] = k
                load[k] += cost(job)
        prev = ops.WGRAD_PENDING
        try:
            for i, (wargs, wkw, dw, _) in enumerate(jobs):
                k = where[i]
                ops.WGRAD_PENDING = self._pends[k]
                with torch.cuda.stream(side[k]):
                    ops.conv_wgrad(*wargs, may_defer=True, dw=dw, **wkw)
        finally:
            ops.WGRAD_PENDING = prev

    def launch(self, wargs, wkw, dw):
        """overlap=True: the kernel goes to ONE side stream the moment autograd reaches the layer (its inputs are ready on the
        current stream) and runs beside the rest of the backward chain; the finishes stay batched at the end."""
        dev = dw.device
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            pool = _WGRAD_STREAMS.setdefault(dev, [])
            if not pool:
                pool.append(torch.cuda.Stream(device=dev))
            self._side = pool[0]
        self._side.wait_stream(main)
        prev = ops.WGRAD_PENDING
        ops.WGRAD_PENDING = self._pend
        try:
            with torch.cuda.stream(self._side):
                ops.conv_wgrad(*wargs, may_defer=True, dw=dw, **wkw)
        finally:
            ops.WGRAD_PENDING = prev

    def __exit__(self, exc_type, exc, tb):
        global _WGRAD_JOBS, _WGRAD_EAGER, _WGRAD_CTX
        jobs = _WGRAD_JOBS
        eager, _WGRAD_EAGER = _WGRAD_EAGER, None
        _WGRAD_CTX = None
        try:
            return self._finish(exc_type, jobs, eager)
        finally:
            _WGRAD_JOBS = None

    def _finish(self, exc_type, jobs, eager):
        if exc_type is not None or not jobs:
            if self._sides:                                              # (an early batch is in flight: join it)
                main = torch.cuda.current_stream(self._sides[0].device)
                for s_ in self._sides:
                    main.wait_stream(s_)
            return False
        dev = jobs[0][2].device
        main = torch.cuda.current_stream(dev)
        if eager is not None:
            try:
                ops.WGRAD_PENDING = self._pend
                with torch.cuda.stream(self._side):
                    ops.conv_wgrad_flush()
            finally:
                ops.WGRAD_PENDING = None
                main.wait_stream(self._side)                             # join (the kernels' inputs are released after it)
                self._pend = []
            return False
        try:
            self.flush_now()                                             # (everything, or what came after the flush point)
            side = self._streams(dev)
            for k, s_ in enumerate(side):
                ops.WGRAD_PENDING = self._pends[k]
                with torch.cuda.stream(s_):
                    ops.conv_wgrad_flush()
        finally:
            ops.WGRAD_PENDING = None
            for s_ in (self._sides or []):
                main.wait_stream(s_)                                     # join (the inputs are released after it)
            self._sides = self._pends = None
        return False
