"""Thin tensor-level wrappers over the C ABI (include/mvster_hip.h).

Every function takes CUDA (=HIP) fp32 tensors, launches on torch's current stream and
returns freshly allocated outputs.  Names and argument meaning follow the reference's
functions in models/mvs4net_utils.py; shapes at this level use the reference's layouts
(NCHW / NCDHW) unless the name says ``_cl`` (channels-last).
"""
import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    """Raw handle of torch's current stream on the current device (the C accessors: ``torch.cuda.current_stream()`` builds
    a Python Stream object per call, ~9 us each and ~200 calls per training step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _chk(t, name):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("mvster_amd.ops.%s: expected a GPU tensor (the HIP path has no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("mvster_amd.ops.%s: the path is fp32-only, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("mvster_amd.ops.%s: tensor must be contiguous" % name)


def relative_projection(proj_matrices):
    """[B,N,2,4,4] -> rt [B,N-1,12] (rot 9 + trans 3).  mvs4net_utils.py:24-26, :1032-1035."""
    pm = proj_matrices.contiguous()
    _chk(pm, "relative_projection")
    B, N = pm.shape[:2]
    rt = torch.empty(B, N - 1, 12, device=pm.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_relative_projection(_ptr(pm), _ptr(rt), B, N, _stream()), "relative_projection")
    return rt


def relative_projection_multi(proj_list):
    """List of per-stage [B,N,2,4,4] tensors -> rt [nstage,B,N-1,12] in one launch."""
    import ctypes
    pms = [p.contiguous() for p in proj_list]
    for p in pms:
        _chk(p, "relative_projection_multi")
    B, N = pms[0].shape[:2]
    rt = torch.empty(len(pms), B, N - 1, 12, device=pms[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * len(pms))(*[p.data_ptr() for p in pms])
    _lib.check(_lib.load().mvster_relative_projection_multi(ctypes.cast(ptrs, ctypes.c_void_p), len(pms), _ptr(rt), B, N,
                                                            _stream()), "relative_projection_multi")
    return rt


def pack_images(imgs):
    """List of N [B,3,H,W] tensors -> [N*B,1,H,W,4] channels-last RGB0 batch (view-major)."""
    import ctypes
    imgs = [i.contiguous() for i in imgs]
    for i in imgs:
        _chk(i, "pack_images")
    B, C, H, W = imgs[0].shape
    if C != 3 or len(imgs) > 16:
        raise RuntimeError("pack_images: expects at most 16 views of [B,3,H,W]")
    out = torch.empty(len(imgs) * B, 1, H, W, 4, device=imgs[0].device, dtype=torch.float32)
    ptrs = (ctypes.c_void_p * len(imgs))(*[i.data_ptr() for i in imgs])
    _lib.check(_lib.load().mvster_pack_images(ctypes.cast(ptrs, ctypes.c_void_p), len(imgs), _ptr(out), B, H, W, _stream()),
               "pack_images")
    return out


def forward_prologue(imgs, proj_list, depth_values, D, h, w, inverse):
    """``pack_images`` + ``relative_projection_multi`` + ``init_range`` (the first stage's hypotheses [B,D,h,w]) in one
    launch -> (packed, rt, hypo); the same bits as the three calls."""
    import ctypes
    imgs = [i.contiguous() for i in imgs]
    pms = [p.contiguous() for p in proj_list]
    dv = depth_values.contiguous()
    for t in imgs + pms + [dv]:
        _chk(t, "forward_prologue")
    B, C, H, W = imgs[0].shape
    if C != 3 or len(imgs) > 16:
        raise RuntimeError("forward_prologue: expects at most 16 views of [B,3,H,W]")
    N = len(imgs)
    dev = imgs[0].device
    # (raw pointers go to the kernel: every shape it derives addresses from is checked here)
    for i, im in enumerate(imgs):
        if tuple(im.shape) != (B, 3, H, W) or im.device != dev:
            raise RuntimeError("forward_prologue: imgs[%d] is %s on %s, expected [%d,3,%d,%d] on %s"
                               % (i, tuple(im.shape), im.device, B, H, W, dev))
    for k, pm in enumerate(pms):
        if tuple(pm.shape) != (B, N, 2, 4, 4) or pm.device != dev:
            raise RuntimeError("forward_prologue: proj_list[%d] is %s, expected [%d,%d,2,4,4]" % (k, tuple(pm.shape), B, N))
    if dv.dim() != 2 or dv.shape[0] != B or dv.shape[1] < 2 or dv.device != dev:
        raise RuntimeError("forward_prologue: depth_values is %s, expected [%d, >= 2]" % (tuple(dv.shape), B))
    if D < 1 or h < 1 or w < 1:
        raise RuntimeError("forward_prologue: bad hypothesis volume %dx%dx%d" % (D, h, w))
    packed = torch.empty(N * B, 1, H, W, 4, device=dev, dtype=torch.float32)
    rt = torch.empty(len(pms), B, N - 1, 12, device=dev, dtype=torch.float32)
    hypo = torch.empty(B, D, h, w, device=dev, dtype=torch.float32)
    ip = (ctypes.c_void_p * N)(*[i.data_ptr() for i in imgs])
    pp = (ctypes.c_void_p * len(pms))(*[p.data_ptr() for p in pms])
    _lib.check(_lib.load().mvster_forward_prologue(ctypes.cast(ip, ctypes.c_void_p), N, _ptr(packed), B, H, W,
                                                   ctypes.cast(pp, ctypes.c_void_p), len(pms), _ptr(rt), _ptr(dv), dv.shape[1],
                                                   _ptr(hypo), D, h, w, int(inverse), _stream()), "forward_prologue")
    return packed, rt, hypo


WGRAD_MAX_SLOTS = 2048        # weight-gradient partial-sum slots per launch (a module attribute: experiments set it)


def to_channels_last(feat_nchw):
    """[B,C,H,W] -> [B,H,W,C] contiguous."""
    return feat_nchw.permute(0, 2, 3, 1).contiguous()


def warp_agg_fwd_cl(ref_cl, src_cl, rt, hypo, G, group_cor=True, attn_fuse_d=True, attn_temp=2.0, want_wsum=False,
                    variant=0):
    """ref_cl [B,h,w,C], src_cl [NV,B,Hs,Ws,C], rt [B,NV,12], hypo [B,D,h,w] ->
    cor_feats channels-last [B,D,h,w,G] (and wsum [B,D,h,w]).  mvs4net_utils.py:1025-1060."""
    for t, n in ((ref_cl, "ref"), (src_cl, "src"), (rt, "rt"), (hypo, "hypo")):
        _chk(t, "warp_agg_fwd:" + n)
    B, h, w, C = ref_cl.shape
    NV, B2, Hs, Ws, C2 = src_cl.shape
    D = hypo.shape[1]
    if B2 != B or C2 != C or tuple(hypo.shape) != (B, D, h, w) or tuple(rt.shape) != (B, NV, 12):
        raise RuntimeError("warp_agg_fwd: inconsistent shapes")
    out = torch.empty(B, D, h, w, G, device=ref_cl.device, dtype=torch.float32)
    wsum = torch.empty(B, D, h, w, device=ref_cl.device, dtype=torch.float32) if want_wsum else None
    rc = _lib.load().mvster_warp_agg_fwd(_ptr(ref_cl), _ptr(src_cl), _ptr(rt), _ptr(hypo), _ptr(out), _ptr(wsum), B, NV,
                                         C, G, D, h, w, Hs, Ws, h * w * C, B * Hs * Ws * C, Hs * Ws * C,
                                         int(group_cor), int(attn_fuse_d), float(attn_temp), int(variant), _stream())
    _lib.check(rc, "warp_agg_fwd")
    return (out, wsum) if want_wsum else out


def warp_agg_fwd_sched_cl(ref_cl, src_cl, rt, G, D, attn_fuse_d=True, attn_temp=2.0, inv_min=None, inv_max=None,
                          depth_values=None):
    """``warp_agg_fwd_cl`` with the stage's hypothesis scheduling in the same launch: the hypotheses are
    ``schedule_inverse_range(inv_min, inv_max)`` of the previous stage's bounds [B,h/2,w/2] (mvs4net_utils.py:79-86) or, with
    ``depth_values`` [B,ndv], ``init_inverse_range`` (:71-77).  -> (cor_feats [B,D,h,w,G], hypo [B,D,h,w]), or None where
    the fused kernel does not apply (the caller runs the two launches; same bits either way)."""
    for t, n in ((ref_cl, "ref"), (src_cl, "src"), (rt, "rt"), (inv_min, "inv_min"), (inv_max, "inv_max"), (depth_values, "depth_values")):
        _chk(t, "warp_agg_fwd_sched:" + n)
    B, h, w, C = ref_cl.shape
    NV, B2, Hs, Ws, C2 = src_cl.shape
    if B2 != B or C2 != C or tuple(rt.shape) != (B, NV, 12):
        raise RuntimeError("warp_agg_fwd_sched: inconsistent shapes")
    lib = _lib.load()
    if not hasattr(lib, "mvster_warp_agg_fwd_sched"):
        return None
    if depth_values is None:
        if inv_min is None or inv_max is None or tuple(inv_min.shape) != (B, h // 2, w // 2) or inv_max.shape != inv_min.shape or (h | w) & 1:
            raise RuntimeError("warp_agg_fwd_sched: the previous stage's bounds must be [B, h/2, w/2]")
        mode, ndv = 1, 0
    else:
        if depth_values.dim() != 2 or depth_values.shape[0] != B or depth_values.shape[1] < 2:
            raise RuntimeError("warp_agg_fwd_sched: depth_values must be [B, >= 2]")
        mode, ndv = 2, depth_values.shape[1]
    out = torch.empty(B, D, h, w, G, device=ref_cl.device, dtype=torch.float32)
    hypo = torch.empty(B, D, h, w, device=ref_cl.device, dtype=torch.float32)
    rc = lib.mvster_warp_agg_fwd_sched(_ptr(ref_cl), _ptr(src_cl), _ptr(rt), _ptr(inv_min), _ptr(inv_max), _ptr(depth_values),
                                       ndv, _ptr(hypo), _ptr(out), None, B, NV, C, G, D, h, w, Hs, Ws, h * w * C,
                                       B * Hs * Ws * C, Hs * Ws * C, int(attn_fuse_d), float(attn_temp), mode, _stream())
    if rc == -3:
        return None
    _lib.check(rc, "warp_agg_fwd_sched")
    return out, hypo


# warp backward: counting-sort the samples by source tile (False / MVSTER_NO_SORTED_SCATTER: scatter windows + atomics)
SORTED_SCATTER = __import__("os").environ.get("MVSTER_NO_SORTED_SCATTER") is None


def warp_agg_bwd_sorted_scratch(B, NV, C, D, h, w, Hs, Ws):
    """(record floats, ints) of scratch ``mvster_warp_agg_bwd_sorted`` needs, or None where it does not apply (source maps
    of more than 2048 tiles of 32 x 32 texels)."""
    import ctypes
    nf, ni = ctypes.c_long(0), ctypes.c_long(0)
    rc = _lib.load().mvster_warp_agg_bwd_sorted_scratch(B, NV, C, D, h, w, Hs, Ws, ctypes.addressof(nf), ctypes.addressof(ni))
    if rc == -3:
        return None
    _lib.check(rc, "warp_agg_bwd_sorted_scratch")
    return nf.value, ni.value


def warp_agg_bwd_cl(ref_cl, src_cl, rt, hypo, out, wsum, grad_out, G, group_cor=True, attn_fuse_d=True, attn_temp=2.0,
                    deterministic=None, into=None, sorted_scatter=None):
    """Gradients of warp_agg_fwd_cl w.r.t. ref_cl and src_cl.  Inside a workgroup the gradients accumulate in 64-bit
    fixed-point LDS counters (integer atomics: 30x cheaper than ds_add_f32 on gfx950, and associative).
    ``deterministic`` (default: off, unless the environment sets MVSTER_BWD_DETERMINISTIC): the workgroups' scatter
    windows are stored densely and summed by a gather pass in fixed order instead of being flushed with global fp32
    atomics, which makes the source gradient bit-reproducible for every tap that falls inside a window (about 20 %
    slower on smooth depth maps, several times slower when neighbouring pixels' hypotheses are unrelated).
    ``into`` = (g_ref, g_src): write into these contiguous buffers instead of allocating; g_src must come zeroed -- except
    with ``sorted_scatter`` (default ``ops.SORTED_SCATTER``; ``mvster_warp_agg_bwd_sorted``: the samples counting-sorted
    by source tile, no global atomics, bit-reproducible, every texel of g_src written exactly once), which ignores what
    the buffer holds.  Where the sorted form does not apply (huge source maps) the call takes the window form."""
    import ctypes
    import os
    grad_out = grad_out.contiguous()
    for t, n in ((ref_cl, "ref"), (src_cl, "src"), (rt, "rt"), (hypo, "hypo"), (out, "out"), (wsum, "wsum"),
                 (grad_out, "grad_out")):
        _chk(t, "warp_agg_bwd:" + n)
    B, h, w, C = ref_cl.shape
    NV, _, Hs, Ws, _ = src_cl.shape
    D = hypo.shape[1]
    lib = _lib.load()
    if sorted_scatter is None:
        sorted_scatter = SORTED_SCATTER
    scratch = warp_agg_bwd_sorted_scratch(B, NV, C, D, h, w, Hs, Ws) if sorted_scatter else None
    if scratch is not None:
        g_ref, g_src = (torch.empty_like(ref_cl), torch.empty_like(src_cl)) if into is None else into
        if g_ref.shape != ref_cl.shape or g_src.shape != src_cl.shape:
            raise ValueError("warp_agg_bwd: `into` buffers must have the shapes of ref_cl and src_cl")
        _chk(g_ref, "warp_agg_bwd:into[0]")
        _chk(g_src, "warp_agg_bwd:into[1]")
        rec = torch.empty(scratch[0], device=ref_cl.device, dtype=torch.float32)
        ints = torch.empty(scratch[1], device=ref_cl.device, dtype=torch.int32)
        rc = lib.mvster_warp_agg_bwd_sorted(_ptr(ref_cl), _ptr(src_cl), _ptr(rt), _ptr(hypo), _ptr(out), _ptr(wsum),
                                            _ptr(grad_out), _ptr(g_ref), _ptr(g_src), _ptr(rec), _ptr(ints), B, NV, C, G, D, h,
                                            w, Hs, Ws, h * w * C, B * Hs * Ws * C, Hs * Ws * C, int(group_cor),
                                            int(attn_fuse_d), float(attn_temp), _stream())
        _lib.check(rc, "warp_agg_bwd_sorted")
        return g_ref, g_src
    if into is None:
        g_ref = torch.empty_like(ref_cl)
        g_src = torch.zeros_like(src_cl)
    else:
        g_ref, g_src = into
        if g_ref.shape != ref_cl.shape or g_src.shape != src_cl.shape:
            raise ValueError("warp_agg_bwd: `into` buffers must have the shapes of ref_cl and src_cl")
        _chk(g_ref, "warp_agg_bwd:into[0]")
        _chk(g_src, "warp_agg_bwd:into[1]")
    if deterministic is None:
        deterministic = bool(os.environ.get("MVSTER_BWD_DETERMINISTIC"))
    nf, ni = ctypes.c_long(0), ctypes.c_long(0)
    _lib.check(lib.mvster_warp_agg_bwd_scratch(B, NV, C, G, D, h, w, int(attn_fuse_d), ctypes.addressof(nf),
                                               ctypes.addressof(ni)), "warp_agg_bwd_scratch")
    windows = torch.empty(nf.value, device=ref_cl.device, dtype=torch.float32) if deterministic else None
    origins = torch.empty(ni.value, device=ref_cl.device, dtype=torch.int32)
    rc = lib.mvster_warp_agg_bwd(_ptr(ref_cl), _ptr(src_cl), _ptr(rt), _ptr(hypo), _ptr(out), _ptr(wsum),
                                 _ptr(grad_out), _ptr(g_ref), _ptr(g_src), _ptr(windows), _ptr(origins), B, NV, C, G, D, h,
                                 w, Hs, Ws, h * w * C, B * Hs * Ws * C, Hs * Ws * C, int(group_cor), int(attn_fuse_d),
                                 float(attn_temp), _stream())
    _lib.check(rc, "warp_agg_bwd")
    return g_ref, g_src


def init_range(depth_values, ndepths, H, W, inverse):
    """mvs4net_utils.py:61-77 -> [B,D,H,W]."""
    dv = depth_values.contiguous()
    _chk(dv, "init_range")
    B, ndv = dv.shape
    out = torch.empty(B, ndepths, H, W, device=dv.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_init_range(_ptr(dv), ndv, _ptr(out), B, ndepths, H, W, int(inverse), _stream()),
               "init_range")
    return out


def schedule_inverse_range(inverse_min_depth, inverse_max_depth, ndepths, H, W):
    """mvs4net_utils.py:79-86: [B,H/2,W/2] x2 -> [B,D,H,W]."""
    a, b = inverse_min_depth.contiguous(), inverse_max_depth.contiguous()
    _chk(a, "schedule_inverse_range")
    _chk(b, "schedule_inverse_range")
    B, hi, wi = a.shape
    if hi != H // 2 or wi != W // 2:
        raise RuntimeError("schedule_inverse_range: previous stage must be at half resolution")
    out = torch.empty(B, ndepths, H, W, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_schedule_inverse_range(_ptr(a), _ptr(b), _ptr(out), B, ndepths, H, W, _stream()),
               "schedule_inverse_range")
    return out


def schedule_range(cur_depth, ndepth, depth_interval_pixel, H, W):
    """mvs4net_utils.py:88-99; ``depth_interval_pixel`` is a [B] device tensor."""
    a = cur_depth.contiguous()
    itv = depth_interval_pixel.to(device=a.device, dtype=torch.float32).contiguous()
    _chk(a, "schedule_range")
    B, hi, wi = a.shape
    if hi != H // 2 or wi != W // 2:
        raise RuntimeError("schedule_range: previous stage must be at half resolution")
    out = torch.empty(B, ndepth, H, W, device=a.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_schedule_range(_ptr(a), _ptr(itv), _ptr(out), B, ndepth, H, W, _stream()),
               "schedule_range")
    return out


def select_depth(hypo, split_itv, inverse_depth, logits=None, feat_cl=None, prob_w=None, prob_b=None,
                 want_logits=False):
    """softmax / argmax / gather / confidence / inverse bounds (mvs4net_utils.py:1068-1088), with the
    1x1x1 ``prob`` head (:900) fused when ``feat_cl`` [B,D,h,w,CF] is given instead of ``logits``."""
    _chk(hypo, "select_depth:hypo")
    B, D, h, w = hypo.shape
    dev = hypo.device
    CF = 0
    if feat_cl is not None:
        _chk(feat_cl, "select_depth:feat")
        CF = feat_cl.shape[-1]
        prob_w = prob_w.reshape(-1).contiguous()
        prob_b = prob_b.reshape(-1).contiguous()
    else:
        logits = logits.contiguous()
        _chk(logits, "select_depth:logits")
    attn = torch.empty(B, D, h, w, device=dev, dtype=torch.float32)
    depth = torch.empty(B, h, w, device=dev, dtype=torch.float32)
    conf = torch.empty(B, h, w, device=dev, dtype=torch.float32)
    imin = torch.empty(B, h, w, device=dev, dtype=torch.float32) if inverse_depth else None
    imax = torch.empty(B, h, w, device=dev, dtype=torch.float32) if inverse_depth else None
    lo = torch.empty(B, D, h, w, device=dev, dtype=torch.float32) if (want_logits and feat_cl is not None) else None
    rc = _lib.load().mvster_select_depth(_ptr(logits), _ptr(feat_cl), _ptr(prob_w), _ptr(prob_b), CF, _ptr(hypo),
                                         _ptr(attn), _ptr(depth), _ptr(conf), _ptr(imin), _ptr(imax), _ptr(lo), B, D, h,
                                         w, float(split_itv), _stream())
    _lib.check(rc, "select_depth")
    out = {"attn_weight": attn, "depth": depth, "conf": conf}
    if inverse_depth:
        out["inverse_min_depth"] = imin
        out["inverse_max_depth"] = imax
    if want_logits:
        out["logits"] = lo if lo is not None else logits
    return out


def select_depth_bwd(attn, gattn, feat_cl, prob_w):
    """Gradients of ``select_depth(..., feat_cl=, prob_w=, prob_b=)`` w.r.t. the feature volume and the head's parameters
    given d L / d attn_weight: (dfeat [B,D,h,w,8], d prob_w [8], d prob_b [1]).  One kernel + one small row sum."""
    gattn = gattn.contiguous()
    for t, n in ((attn, "attn"), (gattn, "gattn"), (feat_cl, "feat"), (prob_w, "prob_w")):
        _chk(t, "select_depth_bwd:" + n)
    B, D, h, w = attn.shape
    CF = feat_cl.shape[-1]
    if tuple(feat_cl.shape) != (B, D, h, w, CF) or tuple(gattn.shape) != (B, D, h, w) or prob_w.numel() != CF:
        raise RuntimeError("select_depth_bwd: inconsistent shapes")
    dfeat = torch.empty_like(feat_cl)
    partial = torch.empty(B * ((h * w + 255) // 256), CF + 1, device=attn.device, dtype=torch.float32)
    rc = _lib.load().mvster_select_depth_bwd(_ptr(attn), _ptr(gattn), _ptr(feat_cl), _ptr(prob_w), _ptr(dfeat), _ptr(partial),
                                             B, D, h, w, CF, _stream())
    _lib.check(rc, "select_depth_bwd")
    sums = partial.sum(0)
    return dfeat, sums[:CF], sums[CF:]


def upsample_bilinear(x, scale):
    """[B,h,w] -> [B,h*scale,w*scale], align_corners=True (mvs4net_utils.py:1077)."""
    x = x.contiguous()
    _chk(x, "upsample_bilinear")
    B, h, w = x.shape
    out = torch.empty(B, h * scale, w * scale, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_upsample_bilinear(_ptr(x), _ptr(out), B, h, w, h * scale, w * scale, _stream()),
               "upsample_bilinear")
    return out


def upsample_bilinear_multi(xs, H, W):
    """List of [B,h_k,w_k] maps -> list of [B,H,W] maps, align_corners=True, one launch (``upsample_bilinear`` per map, same
    bits)."""
    import ctypes
    xs = [x.contiguous() for x in xs]
    for x in xs:
        _chk(x, "upsample_bilinear_multi")
    if not 1 <= len(xs) <= 8:
        raise RuntimeError("upsample_bilinear_multi: 1..8 maps per launch")
    B = xs[0].shape[0]
    for i, x in enumerate(xs):
        if x.dim() != 3 or x.shape[0] != B or x.device != xs[0].device or x.shape[1] < 1 or x.shape[2] < 1:
            raise RuntimeError("upsample_bilinear_multi: map %d is %s on %s, expected [%d,h,w] on %s"
                               % (i, tuple(x.shape), x.device, B, xs[0].device))
    if H < 1 or W < 1:
        raise RuntimeError("upsample_bilinear_multi: bad output size %dx%d" % (H, W))
    outs = [torch.empty(B, H, W, device=x.device, dtype=torch.float32) for x in xs]
    n = len(xs)
    ip = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    op = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    hs = (ctypes.c_int * n)(*[x.shape[1] for x in xs])
    ws = (ctypes.c_int * n)(*[x.shape[2] for x in xs])
    _lib.check(_lib.load().mvster_upsample_bilinear_multi(ctypes.cast(ip, ctypes.c_void_p), ctypes.cast(op, ctypes.c_void_p),
                                                          ctypes.cast(hs, ctypes.c_void_p), ctypes.cast(ws, ctypes.c_void_p),
                                                          n, B, H, W, _stream()), "upsample_bilinear_multi")
    return outs


def fpn_tail_gather(G, vb, H, W, separable=False):
    """G [NB,1,H/2,W/2,9*CO] (or 4-D), vb [9,CO] -> P [NB,1,H,W,CO]; see include/mvster_hip.h."""
    _chk(G, "fpn_tail_gather:G")
    _chk(vb, "fpn_tail_gather:vb")
    NB = G.shape[0]
    CO = vb.shape[1]
    if G.shape[-1] != 9 * CO or G.numel() != NB * (H // 2) * (W // 2) * 9 * CO:
        raise RuntimeError("fpn_tail_gather: inconsistent shapes")
    P = torch.empty(NB, 1, H, W, CO, device=G.device, dtype=torch.float32)
    ws = torch.empty(NB, H, W // 2, 3 * CO, device=G.device, dtype=torch.float32) if separable else None
    _lib.check(_lib.load().mvster_fpn_tail_gather(_ptr(G), _ptr(vb), _ptr(P), _ptr(ws), NB, H, W, CO, _stream()),
               "fpn_tail_gather")
    return P


def fpn_tail_gather_bwd(gP, pitch=None):
    """gP [NB,1,H,W,CO] -> gG [NB,1,H/2,W/2,pitch]: adjoint of fpn_tail_gather w.r.t. G (channels 9*CO.. are zero)."""
    _chk(gP, "fpn_tail_gather_bwd")
    NB, _, H, W, CO = gP.shape
    pitch = 9 * CO if pitch is None else pitch
    gG = torch.empty(NB, 1, H // 2, W // 2, pitch, device=gP.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_fpn_tail_gather_bwd(_ptr(gP), _ptr(gG), NB, H, W, CO, pitch, _stream()),
               "fpn_tail_gather_bwd")
    return gG


def fpn_lateral_up(x, A, bias, q):
    """x [NB,1,H,W,CI], A [CO,CI], bias [CO], q [NB,1,H/2,W/2,CO] -> bias + A x + up2(q) as [NB,1,H,W,CO]."""
    for t, n in ((x, "x"), (A, "A"), (bias, "bias"), (q, "q")):
        _chk(t, "fpn_lateral_up:" + n)
    NB, _, H, W, CI = x.shape
    CO = A.shape[0]
    if tuple(A.shape) != (CO, CI) or bias.numel() != CO or tuple(q.shape) != (NB, 1, H // 2, W // 2, CO):
        raise RuntimeError("fpn_lateral_up: inconsistent shapes")
    out = torch.empty(NB, 1, H, W, CO, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_fpn_lateral_up(_ptr(x), _ptr(A), _ptr(bias), _ptr(q), _ptr(out), NB, H, W, CI, CO,
                                                 _stream()), "fpn_lateral_up")
    return out


def fpn_tail_fused(x, A, bias, q, vb, H, W):
    """fpn_lateral_up + fpn_tail_gather of the finest FPN level in one launch (the 72-channel half-resolution map stays in
    LDS): x [NB,1,H/2,W/2,16], A [72,16], bias [72], q [NB,1,H/4,W/4,72], vb [9,8] -> P [NB,1,H,W,8], or None where the
    fused kernel does not apply (small or odd maps: the two launches cover those)."""
    for t, n in ((x, "x"), (A, "A"), (bias, "bias"), (q, "q"), (vb, "vb")):
        _chk(t, "fpn_tail_fused:" + n)
    NB, _, Hh, Wh, CI = x.shape
    lib = _lib.load()
    if (CI != 16 or tuple(A.shape) != (72, 16) or tuple(vb.shape) != (9, 8) or H % 4 or W % 4 or H < 16 or W < 64
            or not hasattr(lib, "mvster_fpn_tail_fused")):
        return None
    if (Hh, Wh) != (H // 2, W // 2) or bias.numel() != 72 or tuple(q.shape) != (NB, 1, H // 4, W // 4, 72):
        raise RuntimeError("fpn_tail_fused: inconsistent shapes")
    P = torch.empty(NB, 1, H, W, 8, device=x.device, dtype=torch.float32)
    _lib.check(lib.mvster_fpn_tail_fused(_ptr(x), _ptr(A), _ptr(bias), _ptr(q), _ptr(vb), _ptr(P), NB, H, W, CI, _stream()),
               "fpn_tail_fused")
    return P


def mfma_probe(A, Bm):
    """A [16,4] @ B [4,16] on one v_mfma_f32_16x16x4_f32 (layout test hook)."""
    A, Bm = A.contiguous(), Bm.contiguous()
    D = torch.empty(16, 16, device=A.device, dtype=torch.float32)
    _lib.check(_lib.load().mvster_mfma_probe(_ptr(A), _ptr(Bm), _ptr(D), _stream()), "mfma_probe")
    return D


def _wgrad_tiles(c):
    t = (c + 15) // 16
    return 4 if t == 3 else t


def conv_wgrad_shape(x_cl, gy_cl, kernel, co_keep=None, ci_keep=None, mirrored=False):
    """Shape of what ``conv_wgrad`` returns for these arguments."""
    co_keep = gy_cl.shape[-1] if co_keep is None else co_keep
    ci_keep = x_cl.shape[-1] if ci_keep is None else ci_keep
    return ((ci_keep, co_keep) if mirrored else (co_keep, ci_keep)) + tuple(kernel)


def conv_wgrad(x_cl, gy_cl, kernel, stride, padding, co_keep=None, ci_keep=None, mirrored=False, may_defer=False, dw=None):
    """Weight gradient of the channels-last convolution y = conv(x; W[CO,CI,kd,kh,kw], stride, padding):
    x_cl [B,Di,Hi,Wi,CI], gy_cl [B,Do,Ho,Wo,CO] -> dW [co_keep,ci_keep,kd,kh,kw] (the leading channels; default all).
    With the roles of x and gy swapped it is the gradient of a ConvTranspose weight [cin,cout,...].  ``mirrored``: return
    dW with the taps mirrored and the two channel axes exchanged, [ci_keep,co_keep,kd,kh,kw] (the narrow-output form of
    train_ops).  Two launches: the slot kernel and the finish.  ``may_defer``: inside ``train_ops.deferred_wgrad_finish`` the
    finish is left to the batched launch at the end of the backward pass and dW is NOT valid before -- only for gradients
    of leaf parameters, which nothing reads during the pass.  ``dw``: write into this tensor instead of allocating.
    Autograd of models/mvs4net_utils.py:116-123 etc."""
    _chk(x_cl, "conv_wgrad:x")
    _chk(gy_cl, "conv_wgrad:gy")
    B, Di, Hi, Wi, CI = x_cl.shape
    B2, Do, Ho, Wo, CO = gy_cl.shape
    kd, kh, kw = kernel
    if B2 != B:
        raise RuntimeError("conv_wgrad: batch mismatch")
    ntaps = kd * kh * kw
    cop = _wgrad_tiles(CO) * 16
    packed = CI <= 8 and ntaps > 1            # narrow input side: 16/CIP kernel taps share one MFMA N tile
    if packed:
        cip = 4 if CI <= 4 else 8
        tpn = 16 // cip
        ngrp, width = -(-ntaps // tpn), 16
    else:
        cip = _wgrad_tiles(CI) * 16
        ngrp, width = ntaps, cip
    rows = B * Do * Ho
    slot_bytes = ngrp * cop * width * 4
    # workgroups = slots of `partial` (added in a fixed order by the finish kernel: deterministic).  Measured on the
    # config-4 step: 1024 slots 26.0 ms, 512 26.4 ms, 256 27.9 ms -- the kernels want the parallelism more than the
    # reduction minds the size.
    # (at least two workgroups per CU where the layer has the rows for it: the small full-resolution layers -- the depth
    #  head's 4 -> 32 at [2, 256, 320] ran on 128 workgroups, 69 us for 0.4 GFLOP)
    nblk = max(1, min(max((rows + 3) // 4, min(rows, 512)), (32 << 20) // slot_bytes, WGRAD_MAX_SLOTS))
    lib = _lib.load()
    pers = _wgrad_pers_slots(lib, CI, CO, kernel, stride, padding, packed)
    if pers > 0:
        nblk = min(nblk, pers)            # the persistent kernel fills one slot per workgroup; 5x5 s2: one resident round
    partial = torch.empty(nblk, ngrp, cop, width, device=x_cl.device, dtype=torch.float32)
    rc = lib.mvster_conv_wgrad(_ptr(x_cl), _ptr(gy_cl), _ptr(partial), nblk, B, Di, Hi, Wi, CI, Do, Ho, Wo, CO,
                               kd, kh, kw, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                               int(packed), _stream())
    _lib.check(rc, "conv_wgrad")
    co_keep = CO if co_keep is None else co_keep
    ci_keep = CI if ci_keep is None else ci_keep
    if co_keep > CO or ci_keep > CI:
        raise RuntimeError("conv_wgrad: cannot keep more channels than the tensors have")
    shape = (ci_keep, co_keep, kd, kh, kw) if mirrored else (co_keep, ci_keep, kd, kh, kw)
    if dw is None:
        dw = torch.empty(shape, device=x_cl.device, dtype=torch.float32)
    elif tuple(dw.shape) != shape or not dw.is_contiguous() or dw.dtype != torch.float32:
        raise RuntimeError("conv_wgrad: `dw` must be a contiguous float32 tensor of shape %s" % (shape,))
    if WGRAD_PENDING is not None and may_defer:
        # deferred: the caller (train_ops.deferred_wgrad_finish) issues every finish of the backward pass in one launch
        WGRAD_PENDING.append((partial, dw, (nblk, ngrp, cop, width, ntaps, cip if packed else 0, co_keep, ci_keep, int(mirrored),
                                            int(mirrored))))
        # (an alias, not `dw` itself: autograd's AccumulateGrad adopts an incoming gradient only when nobody else holds
        #  the tensor object -- otherwise it stores a CLONE, i.e. a copy of the not yet finished buffer)
        return dw.view(dw.shape)
    rc = lib.mvster_conv_wgrad_finish(_ptr(partial), _ptr(dw), nblk, ngrp, cop, width, ntaps, cip if packed else 0,
                                      co_keep, ci_keep, int(mirrored), int(mirrored), _stream())
    _lib.check(rc, "conv_wgrad_finish")
    return dw


WGRAD_PENDING = None          # a list while weight-gradient finishes are being deferred (train_ops.deferred_wgrad_finish)


def conv_wgrad_flush():
    """Issue the deferred finishes (``WGRAD_PENDING``) as one batched launch per 56 layers and empty the list."""
    import ctypes
    pend = WGRAD_PENDING
    if not pend:
        return

    class Rec(ctypes.Structure):
        _fields_ = [("partial", ctypes.c_void_p), ("dw", ctypes.c_void_p)] + [(n, ctypes.c_int) for n in
                   ("nblk", "ngrp", "cop", "width", "ntaps", "cip", "co_lim", "ci_lim", "swap", "flip")]
    arr = (Rec * len(pend))()
    for r, (partial, dw, ints) in zip(arr, pend):
        r.partial, r.dw = partial.data_ptr(), dw.data_ptr()
        (r.nblk, r.ngrp, r.cop, r.width, r.ntaps, r.cip, r.co_lim, r.ci_lim, r.swap, r.flip) = ints
    rc = _lib.load().mvster_conv_wgrad_finish_batch(ctypes.cast(arr, ctypes.c_void_p), len(pend), _stream())
    _lib.check(rc, "conv_wgrad_finish_batch")
    pend.clear()


def col_sum(x):
    """Sum of a channels-last tensor [..., C] over everything but the channels -> [C]; one launch, deterministic (a
    convolution's bias gradient).  C in {4, 8, 16, 32, 64}."""
    _chk(x, "col_sum")
    C = x.shape[-1]
    rows = x.numel() // C
    nblk = _bn_slots(rows, C, 1)
    if nblk <= 0:
        raise RuntimeError("col_sum: unsupported channel count %d" % C)
    partial = torch.empty(nblk, 2, C, device=x.device, dtype=torch.float32)
    out = torch.empty(C, device=x.device, dtype=torch.float32)
    rc = _lib.load().mvster_col_sum(_ptr(x), _ptr(partial), _ptr(out), _ticket(x.device), rows, C, _stream())
    _lib.check(rc, "col_sum")
    return out


_WGRAD_SLOTS = {}


def _wgrad_pers_slots(lib, CI, CO, kernel, stride, padding, packed):
    key = (CI, CO, kernel, stride, padding, packed)
    n = _WGRAD_SLOTS.get(key)
    if n is None:
        n = _WGRAD_SLOTS[key] = lib.mvster_conv_wgrad_slots(CI, CO, *kernel, *stride, *padding, int(packed))
    return n


_BN_SLOTS = {}


def _bn_slots(rows, C, groups):
    key = (rows, C, groups)
    n = _BN_SLOTS.get(key)
    if n is None:
        n = _BN_SLOTS[key] = _lib.load().mvster_bn_slots(rows, C, groups)
    return n


_BN_TRAIN_SLOTS = {}


def _bn_train_slots(rows, C, groups):
    key = (rows, C, groups)
    n = _BN_TRAIN_SLOTS.get(key)
    if n is None:
        n = _BN_TRAIN_SLOTS[key] = _lib.load().mvster_bn_train_slots(rows, C, groups)
    return n


_TICKETS = {}


def _ticket(dev):
    """Address of four zero int32s on ``dev`` for a one-launch reduction's arrival counter / a fused pass's grid barrier
    (the kernels leave them at zero).  Handed out round-robin from a pool: launches that could overlap (other streams)
    practically never share one."""
    hit = _TICKETS.get(dev)
    if hit is None:
        hit = _TICKETS[dev] = [torch.zeros(4 * 4096, device=dev, dtype=torch.int32), 0]
    hit[1] = (hit[1] + 1) % 4096
    return hit[0].data_ptr() + 16 * hit[1]


_BN_FUSED_OK = {}
# Small tensors: BatchNorm statistics + apply (reduce + apply) in ONE launch with a resident-grid barrier.  OFF: measured on
# the config-4 step it takes 62 launches away (216 -> 154) and saves nothing -- 12.92 against 12.88 ms: the barrier is ~6
# dependent memory round trips (slot drain, ticket, slot loads, statistics publish, flag poll, statistics loads), 13 us
# for a tensor the two launches handle in 10 + 5 (profiles/r06_*bn_fused_ab.txt).  Kept behind this switch with its tests.
BN_FUSED = False


def bn_fused_ok(rows, C, groups, backward):
    key = (rows, C, groups, backward)
    ok = _BN_FUSED_OK.get(key)
    if ok is None:
        ok = _BN_FUSED_OK[key] = bool(_lib.load().mvster_bn_fused_ok(rows, C, groups, int(backward)))
    return ok and BN_FUSED


BN_TAILLESS = True            # training BatchNorm: slot sums in the apply kernels' prologue (False: last-arriver reductions)


def bn_train_fwd(x, weight, bias, running_mean, running_var, eps, momentum, relu, groups=1, num_batches_tracked=None, skip=None):
    """Training-mode BatchNorm (+ ReLU, + skip): slots, then apply with the statistics finished in its prologue -> (y, pack);
    the running statistics / counter are updated in place like ``bn_batch_stats``."""
    _chk(x, "bn_train_fwd:x")
    _chk(skip, "bn_train_fwd:skip")
    C = x.shape[-1]
    rows = x.numel() // C // groups
    if skip is not None and tuple(skip.shape) != tuple(x.shape):
        raise RuntimeError("bn_train_fwd: skip must have the shape of x")
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise RuntimeError("bn_train_fwd: num_batches_tracked must be an int64 tensor on the device")
    nblk = _bn_train_slots(rows, C, groups)
    if nblk <= 0:
        raise RuntimeError("bn_train_fwd: unsupported channel count %d" % C)
    partial = torch.empty(groups, nblk, 2, C, device=x.device, dtype=torch.float32)
    pack = torch.empty(5, groups, C, device=x.device, dtype=torch.float32)
    y = torch.empty_like(x)
    rc = _lib.load().mvster_bn_train_fwd(_ptr(x), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                                         _ptr(num_batches_tracked), _ptr(skip), _ptr(partial), _ptr(y), _ptr(pack), rows, C,
                                         int(relu), int(groups), float(eps), float(momentum), _stream())
    _lib.check(rc, "bn_train_fwd")
    return y, pack


def bn_train_bwd(x, gy, pack, relu, groups=1):
    """Backward of ``bn_train_fwd`` -> (dx, dbeta, dgamma): slots, then dx with the sums formed in its prologue."""
    _chk(x, "bn_train_bwd:x")
    _chk(gy, "bn_train_bwd:gy")
    _chk(pack, "bn_train_bwd:pack")
    C = x.shape[-1]
    rows = x.numel() // C // groups
    nblk = _bn_train_slots(rows, C, groups)
    if nblk <= 0:
        raise RuntimeError("bn_train_bwd: unsupported channel count %d" % C)
    partial = torch.empty(groups, nblk, 2, C, device=x.device, dtype=torch.float32)
    dgb = torch.empty(2, C, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    rc = _lib.load().mvster_bn_train_bwd(_ptr(x), _ptr(gy), _ptr(pack), _ptr(partial), _ptr(dgb[0]), _ptr(dgb[1]), _ptr(dx), rows,
                                         C, int(relu), int(groups), _stream())
    _lib.check(rc, "bn_train_bwd")
    return dx, dgb[1], dgb[0]


def bn_fwd_fused(x, weight, bias, running_mean, running_var, eps, momentum, relu, groups=1, num_batches_tracked=None, skip=None):
    """``bn_batch_stats`` + ``bn_relu_fwd`` of a small tensor in ONE launch -> (y, pack)."""
    _chk(x, "bn_fwd_fused:x")
    _chk(skip, "bn_fwd_fused:skip")
    C = x.shape[-1]
    rows = x.numel() // C // groups
    if skip is not None and tuple(skip.shape) != tuple(x.shape):
        raise RuntimeError("bn_fwd_fused: skip must have the shape of x")
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise RuntimeError("bn_fwd_fused: num_batches_tracked must be an int64 tensor on the device")
    partial = torch.empty(groups, 128, 2, C, device=x.device, dtype=torch.float32)
    pack = torch.empty(5, groups, C, device=x.device, dtype=torch.float32)
    y = torch.empty_like(x)
    rc = _lib.load().mvster_bn_fwd_fused(_ptr(x), _ptr(skip), _ptr(y), _ptr(weight), _ptr(bias), _ptr(running_mean),
                                         _ptr(running_var), _ptr(num_batches_tracked), _ptr(partial), _ptr(pack),
                                         _ticket(x.device), rows, C, int(relu), int(groups), float(eps), float(momentum), _stream())
    _lib.check(rc, "bn_fwd_fused")
    return y, pack


def bn_bwd_fused(x, gy, pack, relu, groups=1):
    """``bn_relu_bwd`` of a small tensor in ONE launch -> (dx, dbeta, dgamma)."""
    _chk(x, "bn_bwd_fused:x")
    _chk(gy, "bn_bwd_fused:gy")
    _chk(pack, "bn_bwd_fused:pack")
    C = x.shape[-1]
    rows = x.numel() // C // groups
    partial = torch.empty(groups, 128, 2, C, device=x.device, dtype=torch.float32)
    sums = torch.empty(groups, 2, C, device=x.device, dtype=torch.float32)
    dgb = torch.empty(2, C, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    rc = _lib.load().mvster_bn_bwd_fused(_ptr(x), _ptr(gy), _ptr(pack), _ptr(partial), _ptr(sums), _ptr(dgb[0]), _ptr(dgb[1]),
                                         _ptr(dx), _ticket(x.device), rows, C, int(relu), int(groups), _stream())
    _lib.check(rc, "bn_bwd_fused")
    return dx, dgb[1], dgb[0]


def bn_batch_stats(x, weight, bias, running_mean, running_var, eps, momentum, groups=1, num_batches_tracked=None):
    """-> pack [5, groups, C] = (mean, biased var, rstd, scale, shift); running_mean / running_var (or None) are
    updated in place, one exponential-average step per group, and num_batches_tracked (or None) += groups."""
    _chk(x, "bn_batch_stats:x")
    C = x.shape[-1]
    rows = x.numel() // C // groups
    lib = _lib.load()
    nblk = _bn_slots(rows, C, groups)
    if nblk <= 0:
        raise RuntimeError("bn_batch_stats: unsupported channel count %d" % C)
    partial = torch.empty(groups, nblk, 2, C, device=x.device, dtype=torch.float32)
    pack = torch.empty(5, groups, C, device=x.device, dtype=torch.float32)
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise RuntimeError("bn_batch_stats: num_batches_tracked must be an int64 tensor on the device")
    rc = lib.mvster_bn_stats(_ptr(x), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                             _ptr(num_batches_tracked), _ptr(partial), _ptr(pack), _ticket(x.device), rows, C, int(groups),
                             float(eps), float(momentum), _stream())
    _lib.check(rc, "bn_stats")
    return pack


def bn_relu_fwd(x, scale, shift, relu, groups=1, skip=None):
    """y = relu(x*scale + shift) (+ skip) on a channels-last tensor [groups*n, ..., C] (training-mode BatchNorm apply);
    scale and shift are [groups, C]: each of the `groups` equal slices along dim 0 has its own statistics; ``skip`` is
    an optional tensor of x's shape added after the activation (the U-Net's skip connections)."""
    _chk(x, "bn_relu_fwd:x")
    C = x.shape[-1]
    if skip is not None:
        _chk(skip, "bn_relu_fwd:skip")
        if tuple(skip.shape) != tuple(x.shape):
            raise RuntimeError("bn_relu_fwd: skip must have the shape of x")
    y = torch.empty_like(x)
    rows = x.numel() // C // groups
    rc = _lib.load().mvster_bn_relu_fwd(_ptr(x), _ptr(scale), _ptr(shift), _ptr(skip), _ptr(y), rows, C, int(relu),
                                        int(groups), _stream())
    _lib.check(rc, "bn_relu_fwd")
    return y


def bn_relu_bwd(x, gy, scale, shift, mean, rstd, relu, groups=1, frozen=False):
    """-> (dx, dbeta [C] = sum g, dgamma [C] = sum g*xh) with g = gy*(y>0), xh = (x-mean)*rstd (BatchNorm + ReLU backward,
    batch statistics per group; the parameter gradients are summed over the groups).  ``frozen``: mean /
    rstd are constants (running statistics), so dx = g * scale."""
    _chk(x, "bn_relu_bwd:x")
    _chk(gy, "bn_relu_bwd:gy")
    C = x.shape[-1]
    rows = x.numel() // C // groups
    lib = _lib.load()
    nblk = _bn_slots(rows, C, groups)
    if nblk <= 0:
        raise RuntimeError("bn_relu_bwd: unsupported channel count %d" % C)
    partial = torch.empty(groups, nblk, 2, C, device=x.device, dtype=torch.float32)
    sums = torch.empty(groups, 2, C, device=x.device, dtype=torch.float32)
    dgb = torch.empty(2, C, device=x.device, dtype=torch.float32)
    rc = lib.mvster_bn_relu_bwd_reduce(_ptr(x), _ptr(gy), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(partial),
                                       _ptr(sums), _ptr(dgb[0]), _ptr(dgb[1]), _ticket(x.device), rows, C, int(relu), int(groups),
                                       _stream())
    _lib.check(rc, "bn_relu_bwd_reduce")
    dx = torch.empty_like(x)
    rc = lib.mvster_bn_relu_bwd_apply(_ptr(x), _ptr(gy), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(sums),
                                      _ptr(dx), rows, C, int(relu), int(groups), int(frozen), _stream())
    _lib.check(rc, "bn_relu_bwd_apply")
    return dx, dgb[1], dgb[0]


def sinkhorn_pixels(attn, hypo, gt, iters, eps, mask=None, continuous=False):
    """Per-pixel Sinkhorn OT loss and its gradient w.r.t. attn: attn, hypo [B,D,H,W], gt [B,H,W] ->
    (loss_pix [B,H,W], jac [B,D,H,W]).  models/mvs4net_utils.py:1096-1142; ``continuous`` selects the
    ot_continous form (:1111-1123), which also reads ``mask`` [B,H,W] (float, > 0.5 = valid)."""
    for t, n in ((attn, "attn"), (hypo, "hypo"), (gt, "gt")):
        _chk(t, "sinkhorn:" + n)
    B, D, H, W = attn.shape
    if tuple(hypo.shape) != (B, D, H, W) or tuple(gt.shape) != (B, H, W):
        raise RuntimeError("sinkhorn: inconsistent shapes")
    loss_pix = torch.empty(B, H, W, device=attn.device, dtype=torch.float32)
    jac = torch.empty_like(attn)
    if continuous:
        _chk(mask, "sinkhorn:mask")
        if mask is None or tuple(mask.shape) != (B, H, W):
            raise RuntimeError("sinkhorn: the continuous form needs mask [B,H,W]")
        rc = _lib.load().mvster_sinkhorn_continuous(_ptr(attn), _ptr(hypo), _ptr(gt), _ptr(mask), _ptr(loss_pix), _ptr(jac), B,
                                                    D, H * W, int(iters), float(eps), _stream())
    else:
        rc = _lib.load().mvster_sinkhorn(_ptr(attn), _ptr(hypo), _ptr(gt), _ptr(loss_pix), _ptr(jac), B, D, H * W, int(iters),
                                         float(eps), _stream())
    _lib.check(rc, "sinkhorn")
    return loss_pix, jac


def stage_loss_terms(hypo, gt, mask, loss_pix, mono=None, inverse_depth=False):
    """The per-pixel terms of one stage of MVS4net_loss around the OT term (models/MVS4Net.py:131-151), one launch:
    hypo [B,D,H,W], gt / mask (float, > 0.5 = valid) / loss_pix [B,H,W], mono [B,H,W] or None -> terms [5, B*H*W] =
    valid, valid*|mono-gt|, valid*out_of_range, valid*loss_pix, valid*sign(mono-gt)."""
    for t, n in ((hypo, "hypo"), (gt, "gt"), (mask, "mask"), (loss_pix, "loss_pix"), (mono, "mono")):
        _chk(t, "stage_loss_terms:" + n)
    B, D, H, W = hypo.shape
    for t in (gt, mask, loss_pix) + (() if mono is None else (mono,)):
        if tuple(t.shape) != (B, H, W):
            raise RuntimeError("stage_loss_terms: inconsistent shapes")
    if D < 3:
        raise RuntimeError("stage_loss_terms: the range interval needs at least 3 hypotheses")
    terms = torch.empty(5, B * H * W, device=hypo.device, dtype=torch.float32)
    rc = _lib.load().mvster_stage_loss_terms(_ptr(hypo), _ptr(gt), _ptr(mask), _ptr(loss_pix), _ptr(mono), _ptr(terms), B, D,
                                             H * W, int(bool(inverse_depth)), _stream())
    _lib.check(rc, "stage_loss_terms")
    return terms


def upsample2x_cl(x, backward=False, mode="bilinear"):
    """x2 up-sampling of a channels-last map [B,h,w,C] -> [B,2h,2w,C], ``mode`` "bilinear" (align_corners=True) or
    "nearest"; ``backward=True`` is the adjoint [B,2h,2w,C] -> [B,h,w,C]."""
    _chk(x, "upsample2x_cl")
    B, H, W, C = x.shape
    if mode not in ("bilinear", "nearest"):
        raise NotImplementedError("upsample2x_cl: mode %r (the path uses bilinear and nearest only)" % (mode,))
    if C % 4:
        raise NotImplementedError("upsample2x_cl: %d channels (the kernels move 4 channels per lane)" % C)
    lib = _lib.load()
    if backward:
        if (H | W) & 1:
            raise RuntimeError("upsample2x_cl: the adjoint needs even sizes")
        out = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=torch.float32)
        if mode == "nearest":
            rc = lib.mvster_upsample2x_nearest_cl(_ptr(x), _ptr(out), B, H // 2, W // 2, C, 1, _stream())
        else:
            rc = lib.mvster_upsample2x_cl_bwd(_ptr(x), _ptr(out), B, H // 2, W // 2, C, _stream())
    else:
        out = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=torch.float32)
        if mode == "nearest":
            rc = lib.mvster_upsample2x_nearest_cl(_ptr(x), _ptr(out), B, H, W, C, 0, _stream())
        else:
            rc = lib.mvster_upsample2x_cl_fwd(_ptr(x), _ptr(out), B, H, W, C, _stream())
    _lib.check(rc, "upsample2x_cl")
    return out


# ---- training-step glue (csrc/train_glue.hip) ---------------------------------------------------------------------------
def stage_loss_fwd(hypo, gt, mask, loss_pix, mono=None, total_in=None, inverse_depth=False, w_l1=0.0, w_ot=1.0, w_stage=1.0):
    """One stage of MVS4net_loss around the OT term (models/MVS4Net.py:126-153): -> (planes [2, B*H*W], out [6] = #valid,
    l1, out-of-range ratio, ot, weighted = w_stage*(w_l1*l1 + w_ot*ot), total = total_in + weighted).  Two launches."""
    for t, n in ((hypo, "hypo"), (gt, "gt"), (mask, "mask"), (loss_pix, "loss_pix"), (mono, "mono"), (total_in, "total_in")):
        _chk(t, "stage_loss_fwd:" + n)
    B, D, H, W = hypo.shape
    for t in (gt, mask, loss_pix) + (() if mono is None else (mono,)):
        if tuple(t.shape) != (B, H, W):
            raise RuntimeError("stage_loss_fwd: inconsistent shapes")
    if D < 3:
        raise RuntimeError("stage_loss_fwd: the range interval needs at least 3 hypotheses")
    if total_in is not None and total_in.numel() != 1:
        raise RuntimeError("stage_loss_fwd: total_in must be a scalar")
    lib = _lib.load()
    n = B * H * W
    planes = torch.empty(2, n, device=hypo.device, dtype=torch.float32)
    partial = torch.empty(lib.mvster_stage_loss_slots(n), 4, device=hypo.device, dtype=torch.float32)
    out = torch.empty(6, device=hypo.device, dtype=torch.float32)
    rc = lib.mvster_stage_loss_fwd(_ptr(hypo), _ptr(gt), _ptr(mask), _ptr(loss_pix), _ptr(mono), _ptr(total_in), _ptr(planes),
                                   _ptr(partial), _ptr(out), B, D, H * W, int(bool(inverse_depth)), float(w_l1), float(w_ot),
                                   float(w_stage), _stream())
    _lib.check(rc, "stage_loss_fwd")
    return planes, out


def stage_loss_bwd(jac, planes, out, g_total, g_l1, g_ot, w_l1, w_ot, want_attn=True, want_mono=False):
    """Backward of ``stage_loss_fwd``: jac [B,D,H,W]; g_* device scalars or None -> (g_attn [B,D,H,W] | None, g_mono [B,H,W]
    | None).  One launch."""
    B, D, H, W = jac.shape
    for t, n in ((jac, "jac"), (planes, "planes"), (out, "out"), (g_total, "g_total"), (g_l1, "g_l1"), (g_ot, "g_ot")):
        _chk(t, "stage_loss_bwd:" + n)
    if planes.numel() != 2 * B * H * W or out.numel() != 6:
        raise RuntimeError("stage_loss_bwd: inconsistent shapes")
    g_attn = torch.empty_like(jac) if want_attn else None
    g_mono = torch.empty(B, H, W, device=jac.device, dtype=torch.float32) if want_mono else None
    rc = _lib.load().mvster_stage_loss_bwd(_ptr(jac), _ptr(planes), _ptr(out), _ptr(g_total), _ptr(g_l1), _ptr(g_ot), float(w_l1),
                                           float(w_ot), _ptr(g_attn), _ptr(g_mono), B, D, H * W, _stream())
    _lib.check(rc, "stage_loss_bwd")
    return g_attn, g_mono


def mono_depth_fwd(z, dmin, dmax):
    """z [B,...] (H*W values per sample), dmin / dmax [B] -> (depth, sigmoid(z)), both of z's shape
    (models/mvs4net_utils.py:858-866)."""
    for t, n in ((z, "z"), (dmin, "dmin"), (dmax, "dmax")):
        _chk(t, "mono_depth_fwd:" + n)
    B = z.shape[0]
    if dmin.numel() != B or dmax.numel() != B:
        raise RuntimeError("mono_depth_fwd: d_min / d_max must hold one value per sample")
    depth, sig = torch.empty_like(z), torch.empty_like(z)
    rc = _lib.load().mvster_mono_depth_fwd(_ptr(z), _ptr(dmin), _ptr(dmax), _ptr(depth), _ptr(sig), B, z.numel() // B, _stream())
    _lib.check(rc, "mono_depth_fwd")
    return depth, sig


def mono_depth_bwd(g, depth, sig, dmin, dmax):
    for t, n in ((g, "g"), (depth, "depth"), (sig, "sig"), (dmin, "dmin"), (dmax, "dmax")):
        _chk(t, "mono_depth_bwd:" + n)
    B = depth.shape[0]
    if g.numel() != depth.numel() or sig.numel() != depth.numel() or dmin.numel() != B or dmax.numel() != B:
        raise RuntimeError("mono_depth_bwd: inconsistent shapes")
    gz = torch.empty_like(depth)
    rc = _lib.load().mvster_mono_depth_bwd(_ptr(g), _ptr(depth), _ptr(sig), _ptr(dmin), _ptr(dmax), _ptr(gz), B,
                                           depth.numel() // B, _stream())
    _lib.check(rc, "mono_depth_bwd")
    return gz


def upcat(a, b, backward=None):
    """concat(nearest x2 of a [NB,1,H/2,W/2,Ca], b [NB,1,H,W,Cb]) along the channels, one launch; ``backward`` = the
    gradient of that [NB,1,H,W,Ca+Cb] map -> (ga, gb) of a's and b's shapes (a, b: shape donors only)."""
    NB, _, H, W, Cb = b.shape
    Ca = a.shape[-1]
    if tuple(a.shape) != (NB, 1, H // 2, W // 2, Ca) or (H | W) & 1:
        raise RuntimeError("upcat: a must be [NB,1,H/2,W/2,Ca] for b [NB,1,H,W,Cb] with even H, W")
    if (Ca | Cb) & 3:
        raise NotImplementedError("upcat: channel counts %d + %d (the kernel moves 4 channels per lane)" % (Ca, Cb))
    lib = _lib.load()
    if backward is None:
        _chk(a, "upcat:a")
        _chk(b, "upcat:b")
        out = torch.empty(NB, 1, H, W, Ca + Cb, device=b.device, dtype=torch.float32)
        _lib.check(lib.mvster_upcat_fwd(_ptr(a), _ptr(b), _ptr(out), NB, H, W, Ca, Cb, _stream()), "upcat_fwd")
        return out
    _chk(backward, "upcat:g")
    if tuple(backward.shape) != (NB, 1, H, W, Ca + Cb):
        raise RuntimeError("upcat: gradient shape %s" % (tuple(backward.shape),))
    ga = torch.empty(NB, 1, H // 2, W // 2, Ca, device=b.device, dtype=torch.float32)
    gb = torch.empty(NB, 1, H, W, Cb, device=b.device, dtype=torch.float32)
    _lib.check(lib.mvster_upcat_bwd(_ptr(backward), _ptr(ga), _ptr(gb), NB, H, W, Ca, Cb, _stream()), "upcat_bwd")
    return ga, gb


def fine_weights_fwd(wo, wi, bi):
    """wo [CO,CM,3,3], wi [CM,CI(,1,1)], bi [CM] -> (wg [9*CO,CM,1,1], wc [CO,CI,3,3], vb [9,CO]): the composed weights
    of the re-associated finest FPN level (train_ops.fpn_fine_level)."""
    for t, n in ((wo, "wo"), (wi, "wi"), (bi, "bi")):
        _chk(t, "fine_weights_fwd:" + n)
    CO, CM = wo.shape[0], wo.shape[1]
    CI = wi.shape[1]
    if tuple(wo.shape) != (CO, CM, 3, 3) or wi.numel() != CM * CI or bi.numel() != CM:
        raise RuntimeError("fine_weights_fwd: inconsistent shapes")
    dev = wo.device
    wg = torch.empty(9 * CO, CM, 1, 1, device=dev, dtype=torch.float32)
    wc = torch.empty(CO, CI, 3, 3, device=dev, dtype=torch.float32)
    vb = torch.empty(9, CO, device=dev, dtype=torch.float32)
    _lib.check(_lib.load().mvster_fine_weights_fwd(_ptr(wo), _ptr(wi), _ptr(bi), _ptr(wg), _ptr(wc), _ptr(vb), CO, CM, CI,
                                                   _stream()), "fine_weights_fwd")
    return wg, wc, vb


def fine_weights_bwd(wo, wi, bi, g_wg, g_wc, g_vb):
    for t, n in ((wo, "wo"), (wi, "wi"), (bi, "bi"), (g_wg, "g_wg"), (g_wc, "g_wc"), (g_vb, "g_vb")):
        _chk(t, "fine_weights_bwd:" + n)
    CO, CM = wo.shape[0], wo.shape[1]
    CI = wi.shape[1]
    g_wo, g_wi, g_bi = torch.empty_like(wo), torch.empty_like(wi), torch.empty_like(bi)
    _lib.check(_lib.load().mvster_fine_weights_bwd(_ptr(wo), _ptr(wi), _ptr(bi), _ptr(g_wg), _ptr(g_wc), _ptr(g_vb), _ptr(g_wo),
                                                   _ptr(g_wi), _ptr(g_bi), CO, CM, CI, _stream()), "fine_weights_bwd")
    return g_wo, g_wi, g_bi
