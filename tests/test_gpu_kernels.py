"""HIP kernels (through the C ABI) against the CPU oracle / golden vectors.  Needs an MI355X."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mvster_amd import conv_plan as cp
from mvster_amd import modules as M
from mvster_amd import ops
from mvster_amd.synthetic import make_inputs, randomize_state
from oracle import mvs4_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = {}


def _probe_build():
    from mvster_amd import _lib
    return torch.cuda.is_available() and _lib.has_probes()


# The kernel forms kept for the record only (warp variants 4 / 5, convolution variant 7) live in the probe library
# (make -C mvster_amd/csrc probes); their tests run when that library is the one loaded:
#     MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_probes.so python -m pytest tests/test_gpu_kernels.py -m gpu -k "probe or variants or pingpong or window"
PROBES = _probe_build()
needs_probes = pytest.mark.skipif(not PROBES, reason="kernel form kept for the record: needs the probe library (MVSTER_LIB=...libmvster_hip_probes.so)")


def test_product_library_has_no_probe_forms():
    """The shipped library reads no environment switch and refuses the kernel forms kept for the record."""
    from mvster_amd import _lib
    if PROBES:
        pytest.skip("probe library loaded")
    assert _lib.load().mvster_build_flags() == 0
    ref = torch.randn(1, 8, 16, 8, device=DEV)
    src = torch.randn(2, 1, 8, 16, 8, device=DEV)
    rt = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], device=DEV).view(1, 1, 12).repeat(1, 2, 1).contiguous()
    hypo = 500 + torch.rand(1, 4, 8, 16, device=DEV)
    for variant in (4, 5):
        with pytest.raises(RuntimeError):
            ops.warp_agg_fwd_cl(ref, src, rt, hypo, 4, True, True, 2.0, variant=variant)


def note(name, **kv):
    REPORT[name] = {k: (float(v) if not isinstance(v, (list, str)) else v) for k, v in kv.items()}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_kernels.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def cl5(x):   # NCDHW -> NDHWC
    return x.permute(0, 2, 3, 4, 1).contiguous()


def twin(oracle_module, m):
    """The oracle's PyTorch module with the product module's parameters: the CPU reference."""
    oracle_module.load_state_dict(m.state_dict(), strict=True)
    return oracle_module.eval()


def test_library_is_loaded_from_the_tree():
    from mvster_amd import _lib
    lib = _lib.load()
    assert os.path.samefile(_lib.LIB_PATH, os.path.join(os.path.dirname(_lib.__file__), "csrc", "libmvster_hip.so"))
    assert lib.mvster_warp_agg_fwd is not None


def test_mfma_fragment_layout():
    """Asymmetric operands: a transposed A, B or D mapping cannot pass."""
    g = torch.Generator().manual_seed(0)
    A = torch.randn(16, 4, generator=g)
    B = torch.randn(4, 16, generator=g)
    D = ops.mfma_probe(A.to(DEV), B.to(DEV)).cpu()
    want = (A.double() @ B.double()).float()
    err = (D - want).abs().max().item()
    note("mfma_probe", max_abs=err)
    assert err <= 1e-6


def test_relative_projection():
    _, proj, _ = make_inputs(5, 512, 640, seed=0, batch=2)
    for stage in ("stage1", "stage4"):
        pm = proj[stage]
        rt = ops.relative_projection(pm.to(DEV)).cpu()
        Pr = O.compose_projection(pm[:, 0]).double()
        for v in range(1, 5):
            P = torch.matmul(O.compose_projection(pm[:, v]).double(), torch.inverse(Pr))
            want = torch.cat([P[:, :3, :3].reshape(2, 9), P[:, :3, 3]], 1)
            got = rt[:, v - 1].double()
            assert torch.all((got - want).abs() <= 5e-7 * want.abs().clamp(min=1.0)), (stage, v)


def test_pack_images_and_multi_projection():
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(2, 3, 12, 20, generator=g) for _ in range(3)]
    out = ops.pack_images([i.to(DEV) for i in imgs]).cpu()
    want = torch.zeros(6, 1, 12, 20, 4)
    want[:, 0, :, :, :3] = torch.stack(imgs, 0).reshape(6, 3, 12, 20).permute(0, 2, 3, 1)
    assert torch.equal(out, want)
    _, proj, _ = make_inputs(4, 128, 192, seed=3, batch=2)
    names = ["stage%d" % s for s in range(1, 5)]
    multi = ops.relative_projection_multi([proj[n].to(DEV) for n in names]).cpu()
    for s, n in enumerate(names):
        assert torch.equal(multi[s], ops.relative_projection(proj[n].to(DEV)).cpu())


@pytest.mark.parametrize("inverse", [True, False])
def test_forward_prologue_equals_its_three_launches(inverse):
    """mvster_forward_prologue = pack_images + relative_projection_multi + init_range, bit for bit; odd sizes, 1..4 stages."""
    g = torch.Generator().manual_seed(1)
    for (B, N, H, W, D, h, w, nst) in ((2, 3, 12, 20, 8, 3, 5, 4), (1, 5, 64, 128, 8, 8, 16, 4), (3, 2, 9, 7, 5, 9, 7, 1)):
        imgs = [torch.rand(B, 3, H, W, generator=g).to(DEV) for _ in range(N)]
        _, proj, _ = make_inputs(N, 128, 192, seed=3, batch=B)
        pms = [proj["stage%d" % (s + 1)].to(DEV) for s in range(nst)]
        dv = (torch.rand(B, 3, generator=g).sort(1)[0] * 500 + 400).to(DEV)
        packed, rt, hypo = ops.forward_prologue(imgs, pms, dv, D, h, w, inverse)
        assert torch.equal(packed, ops.pack_images(imgs))
        assert torch.equal(rt, ops.relative_projection_multi(pms))
        assert torch.equal(hypo, ops.init_range(dv, D, h, w, inverse=inverse))
    with pytest.raises(RuntimeError, match="bad shape"):                     # more hypothesis pixels than image pixels
        ops.forward_prologue(imgs, pms, dv, D, 64, 64, inverse)


def test_upsample_bilinear_multi_equals_single_launches():
    g = torch.Generator().manual_seed(2)
    xs = [torch.rand(2, 8 * k, 12 * k, generator=g).to(DEV) for k in (1, 2, 4)]
    got = ops.upsample_bilinear_multi(xs, 64, 96)
    for x, o, k in zip(xs, got, (8, 4, 2)):
        assert o.shape == (2, 64, 96) and torch.equal(o, ops.upsample_bilinear(x, k))
    one = ops.upsample_bilinear_multi(xs[:1], 8, 12)                          # x1: the identity, exactly
    assert torch.equal(one[0], xs[0])
    with pytest.raises(RuntimeError, match="1..8 maps"):
        ops.upsample_bilinear_multi([], 8, 8)


def test_schedulers(golden):
    g = golden("g5_sched")
    dv = g.t("dv", DEV)
    assert torch.equal(ops.init_range(dv, 8, 6, 10, inverse=True).cpu(), g.t("init_inverse_8"))
    assert torch.equal(ops.init_range(dv, 8, 6, 10, inverse=False).cpu(), g.t("init_range_8"))
    worst = 0.0
    for D in (8, 4):
        out = ops.schedule_inverse_range(g.t("inv_min", DEV), g.t("inv_max", DEV), D, 24, 40).cpu()
        want = g.t("sched_inverse_%d" % D)
        worst = max(worst, ((out - want).abs() / want.abs()).max().item())
    out = ops.schedule_range(g.t("cur_depth", DEV), 4, g.t("itv", DEV), 24, 40).cpu()
    want = g.t("sched_range_4")
    worst = max(worst, ((out - want).abs() / want.abs()).max().item())
    note("schedulers", max_rel=worst)
    assert worst <= 5e-7          # <= 3 ulp: ATen contracts its lerp into FMAs differently


@pytest.mark.parametrize("name", ["d8_s0", "d4_s2", "d4_s3_ties", "d8_s1_b2"])
def test_select_depth(golden, name):
    g = golden("g4_select")
    s = int(g.np(name + "_stage_idx"))
    r = ops.select_depth(g.t(name + "_hypo", DEV), 0.5, True, logits=g.t(name + "_logits", DEV))
    attn = r["attn_weight"].cpu()
    want_attn = g.t(name + "_attn_weight")
    assert (attn - want_attn).abs().max() <= 3e-7
    top2 = want_attn.topk(2, dim=1)[0]
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-6) | (top2[:, 0] == top2[:, 1])     # exact ties must agree too
    assert torch.equal(r["depth"].cpu()[clear], g.t(name + "_depth")[clear])
    assert (r["inverse_min_depth"].cpu() - g.t(name + "_inverse_min_depth"))[clear].abs().max() <= 1e-9
    assert (r["inverse_max_depth"].cpu() - g.t(name + "_inverse_max_depth"))[clear].abs().max() <= 1e-9
    conf = ops.upsample_bilinear(r["conf"], 2 ** (3 - s)).cpu()
    assert (conf - g.t(name + "_photometric_confidence")).abs().max() <= 5e-7


def test_select_depth_fused_prob():
    g = torch.Generator().manual_seed(1)
    B, D, h, w, CF = 2, 4, 9, 13, 8
    feat = torch.randn(B, D, h, w, CF, generator=g)
    pw, pb = torch.randn(CF, generator=g), torch.randn(1, generator=g)
    hypo = 500 + 100 * torch.rand(B, D, h, w, generator=g)
    r = ops.select_depth(hypo.to(DEV), 1.0, True, feat_cl=feat.to(DEV), prob_w=pw.to(DEV), prob_b=pb.to(DEV),
                         want_logits=True)
    logits = feat @ pw + pb
    assert (r["logits"].cpu() - logits).abs().max() <= 1e-5
    want = O.select_depth(r["logits"].cpu(), hypo, 3, True, 1.0)
    assert (r["attn_weight"].cpu() - want["attn_weight"]).abs().max() <= 3e-7
    assert torch.equal(r["depth"].cpu(), want["depth"])


def _oracle_rt(pm):
    ref_p = O.compose_projection(pm[:, 0])
    rts = []
    for v in range(1, pm.shape[1]):
        P = torch.matmul(O.compose_projection(pm[:, v]), torch.inverse(ref_p))
        rts.append(torch.cat([P[:, :3, :3].reshape(-1, 9), P[:, :3, 3]], 1))
    return torch.stack(rts, 1).contiguous()


@pytest.mark.parametrize("name", ["gc_t2", "gc_t1", "sq_t2", "gc_nofuse", "gc_b2"])
def test_warp_agg_forward(golden, name):
    """Fused warp + correlation + attention aggregation vs the reference's captured cor_feats."""
    g = golden("g2_aggregate")
    n, C, G, D, gc, fuse, temp = g.np(name + "_cfg").tolist()
    C, G, gc, fuse = int(C), int(G), bool(gc), bool(fuse)
    feats = g.t(name + "_feats")                      # [N,B,C,h,w]
    pm, hypo, want = g.t(name + "_proj"), g.t(name + "_hypo"), g.t(name + "_cor")
    f_cl = feats.permute(0, 1, 3, 4, 2).contiguous().to(DEV)
    Gk = G if gc else C
    # (1) with the reference's own fp32 relative projection: pure per-pixel arithmetic
    out = ops.warp_agg_fwd_cl(f_cl[0], f_cl[1:], _oracle_rt(pm).to(DEV), hypo.to(DEV), Gk, gc, fuse, temp, variant=1)
    got = out.permute(0, 4, 1, 2, 3).cpu()
    tight = (got - want).abs().max().item()
    # the lane-split, wave-local and pixel-major kernels are bit-identical to the one-thread-per-(pixel, d) form
    for variant in (0, 2, 3) + ((4,) if PROBES else ()):            # (4 = pixel-major: probe library only)
        o2 = ops.warp_agg_fwd_cl(f_cl[0], f_cl[1:], _oracle_rt(pm).to(DEV), hypo.to(DEV), Gk, gc, fuse, temp,
                                 variant=variant)
        assert torch.equal(o2, out), (name, variant)
    # (2) with the kernel-computed (fp64-inverse) projection: differs by the fp32 LAPACK noise of the reference
    out2 = ops.warp_agg_fwd_cl(f_cl[0], f_cl[1:], ops.relative_projection(pm.to(DEV)), hypo.to(DEV), Gk, gc, fuse, temp)
    got2 = out2.permute(0, 4, 1, 2, 3).cpu()
    loose = (got2 - want).abs().max().item()
    note("warp_agg_" + name, tight_max_abs=tight, own_rt_max_abs=loose, ref_absmax=want.abs().max().item(),
         own_rt_mean_abs=(got2 - want).abs().mean().item())
    scale = want.abs().max().item()
    # measured: tight <= 7e-7, own projection <= 1.9e-5 max / 6e-7 mean (the reference's fp32 LAPACK noise)
    assert tight <= 2e-6 * max(scale, 1.0)
    assert loose <= 1e-4 * max(scale, 1.0)
    assert (got2 - want).abs().mean().item() <= 3e-6 * max(scale, 1.0)


@pytest.mark.parametrize("C,G,gc,D,fuse", [(64, 8, True, 48, True), (32, 8, True, 32, True), (16, 4, True, 17, False),
                                           (8, 4, True, 64, True), (16, 16, False, 40, True), (8, 8, False, 24, False),
                                           (32, 4, True, 12, True)])
def test_warp_agg_forward_any_hypothesis_count(C, G, gc, D, fuse):
    """More hypotheses per stage than the shipped cascade's 8 / 4 (a free --ndepths of the reference): the general kernel with
    64 / 32 / 16 pixels x D hypotheses per workgroup against the oracle's restatement of mvs4net_utils.py:1015-1060 fed with
    the oracle's own relative projection (ragged pixel counts, two batch items).  Bound: these inputs draw every pixel's
    hypotheses from the whole depth range (disparities of hundreds of pixels) and unit-variance features, which puts the fp32
    coordinate arithmetic of grid_sample and of the kernel 4-7e-6 apart relative to the output scale -- the same on the
    D = 12 case, which runs the form the golden cases pin at 7e-7 on realistic hypotheses."""
    from mvster_amd.synthetic import make_inputs as mk
    h, w, nv, B = 21, 37, 3, 2
    g = torch.Generator().manual_seed(C + D)
    _, proj, dv = mk(nviews=nv + 1, H=h * 8, W=w * 8, batch=B, seed=D, rotate=True)
    pm = proj["stage1"]
    feats = torch.randn(nv + 1, B, C, h, w, generator=g)
    hypo = dv[:, :1, None, None] + (dv[:, -1:, None, None] - dv[:, :1, None, None]) * torch.rand(B, D, h, w, generator=g)
    with torch.no_grad():
        want = O.aggregate_views(list(feats), pm, hypo, gc, G, attn_temp=2.0, attn_fuse_d=fuse)
    f_cl = feats.permute(0, 1, 3, 4, 2).contiguous().to(DEV)
    out = ops.warp_agg_fwd_cl(f_cl[0], f_cl[1:], _oracle_rt(pm).to(DEV), hypo.to(DEV), G, gc, fuse, 2.0)
    from mvster_amd import _lib
    form = "8" if D <= 8 else "16" if D <= 16 else "32, 32" if D <= 32 else "64, 16"
    assert _lib.last_kernel() == "warp_agg_fwd_kernel<%d, %d, %s, %s>" % (C, G, "true" if gc else "false", form)
    got = out.permute(0, 4, 1, 2, 3).cpu()
    err = (got - want).abs().max().item()
    note("warp_agg_D%d_C%d" % (D, C), max_abs=err, ref_absmax=want.abs().max().item())
    assert err <= 1.2e-5 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize("D,inverse", [(17, True), (48, True), (200, False)])
def test_select_depth_any_hypothesis_count(D, inverse):
    """select_depth beyond the register form's 16 hypotheses (mv::select_pixel_any): against the oracle's restatement of
    mvs4net_utils.py:1068-1088, exact ties included; with and without the fused prob head."""
    g = torch.Generator().manual_seed(D)
    B, h, w, CF = 2, 11, 19, 8
    hypo = (400 + 300 * torch.rand(B, D, h, w, generator=g)).sort(1)[0]
    logits = 2 * torch.randn(B, D, h, w, generator=g)
    logits[:, 5, :2] = logits[:, 3, :2] = logits.max(1)[0][:, :2] + 1           # exact ties: the first maximum wins
    r = ops.select_depth(hypo.to(DEV), 0.5, inverse, logits=logits.to(DEV))
    want = O.select_depth(logits, hypo, 3, inverse, 0.5)
    assert (r["attn_weight"].cpu() - want["attn_weight"]).abs().max() <= 3e-7
    top2 = want["attn_weight"].topk(2, dim=1)[0]
    clear = ((top2[:, 0] - top2[:, 1]) > 1e-6) | (top2[:, 0] == top2[:, 1])
    assert torch.equal(r["depth"].cpu()[clear], want["depth"][clear])
    assert torch.equal(r["depth"].cpu()[:, :2], hypo[:, 3, :2])
    if inverse:
        assert (r["inverse_min_depth"].cpu() - want["inverse_min_depth"])[clear].abs().max() <= 1e-9
    feat = torch.randn(B, D, h, w, CF, generator=g)
    pw, pb = torch.randn(CF, generator=g), torch.randn(1, generator=g)
    r2 = ops.select_depth(hypo.to(DEV), 0.5, inverse, feat_cl=feat.to(DEV), prob_w=pw.to(DEV), prob_b=pb.to(DEV), want_logits=True)
    assert (r2["logits"].cpu() - (feat @ pw + pb)).abs().max() <= 1e-5
    want2 = O.select_depth(r2["logits"].cpu(), hypo, 3, inverse, 0.5)
    assert (r2["attn_weight"].cpu() - want2["attn_weight"]).abs().max() <= 3e-7


@pytest.mark.parametrize("C,G,D", [(64, 8, 8), (32, 8, 8), (16, 4, 4), (8, 4, 4), (8, 8, 8), (16, 8, 4), (32, 4, 8),
                                   (64, 8, 4), (8, 4, 8)])
@pytest.mark.parametrize("fuse", [True, False])
def test_warp_agg_kernel_variants_bit_identical(C, G, D, fuse):
    """Every launch form of the fused kernel (one thread per (pixel, d); workgroup-level lane split; wave-local;
    pixel-major with packed hypothesis pairs and buffer loads; the default choice) returns the same bits, incl. ragged pixel counts,
    out-of-image taps, odd view counts and the saved softmax mass."""
    from mvster_amd.synthetic import make_inputs as mk
    for (h, w, nv, B) in ((37, 53, 3, 2), (16, 24, 4, 1), (5, 7, 2, 1)):
        g = torch.Generator().manual_seed(C * 131 + D * 7 + h)
        _, proj, dv = mk(nviews=nv + 1, H=h * 8, W=w * 8, batch=B, seed=h, rotate=True)
        pm = proj["stage1"]
        ref = torch.randn(B, h, w, C, generator=g)
        src = torch.randn(nv, B, h, w, C, generator=g)
        hypo = dv[:, :1, None, None] + (dv[:, -1:, None, None] - dv[:, :1, None, None]) * torch.rand(B, D, h, w, generator=g)
        rt = ops.relative_projection(pm.to(DEV))
        args = (ref.to(DEV), src.to(DEV), rt, hypo.to(DEV), G, True, fuse, 2.0)
        base, wbase = ops.warp_agg_fwd_cl(*args, want_wsum=True, variant=1)
        assert torch.isfinite(base).all() and base.abs().max() > 0
        for variant in (0, 2, 3) + ((4,) if PROBES else ()):
            o, ws = ops.warp_agg_fwd_cl(*args, want_wsum=True, variant=variant)
            assert torch.equal(o, base), (C, G, D, h, w, variant)
            assert torch.equal(ws, wbase), (C, G, D, h, w, variant)


@pytest.mark.parametrize("D,inverse,G,shape", [(4, True, 4, (1, 37, 53)), (8, True, 8, (2, 16, 24)), (4, False, 4, (1, 64, 80)),
                                               (3, True, 4, (1, 9, 70)), (16, True, 8, (1, 10, 12)), (2, False, 4, (1, 6, 6)),
                                               (4, True, 4, (2, 72, 200)), (8, True, 8, (1, 40, 136)), (4, True, 4, (1, 256, 320))])
def test_fused_conv11_selection_bit_identical(D, inverse, G, shape):
    """Reg2dPlan.select: reg2d's last layer + prob head + softmax / argmax / gather / bounds in one launch
    (mvster_deconv_select) against the two launches (deconv_small with the fused prob head, then select_depth): every
    output equal, incl. the optional logits, odd sizes and every D the selection kernel takes.  D = 4 and 8 (the shipped
    cascade) run the MFMA form on the persistent LDS-DMA ring (deconv_select.hip: two- and four-row tiles, ragged tiles,
    several tiles per workgroup, B = 2), whose K order reproduces the VALU kernel's FMA chain: the same bits."""
    torch.manual_seed(D * 7 + G)
    m = M.reg2d(input_channel=G, base_channel=8)
    m.load_state_dict(randomize_state(m.state_dict(), seed=3, prob_gain=8.0))
    plan = cp.Reg2dPlan(m.to(DEV).eval())
    B, h, w = shape
    h, w = h // 8 * 8 or 8, w // 8 * 8 or 8                                  # the U-Net halves three times
    x = torch.randn(B, D, h, w, G, device=DEV)
    hypo = (1.0 / (1.0 / 900 + 2e-5 * (torch.arange(D).view(1, D, 1, 1) + 0.1 * torch.rand(B, D, h, w)))).float().to(DEV)
    cp.FUSE_SELECT = False
    try:
        want = plan.select(x, hypo, 0.5, inverse, want_logits=True)
    finally:
        cp.FUSE_SELECT = True
    got = plan.select(x, hypo, 0.5, inverse, want_logits=True)
    from mvster_amd import _lib
    assert _lib.last_kernel().startswith("deconv_select_mfma_kernel<%d, " % D if D in (4, 8) else "deconv_select_kernel<16>")
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), (k, (got[k] - want[k]).abs().max().item())
    plain = plan.select(x, hypo, 0.5, inverse)
    assert "logits" not in plain and torch.equal(plain["depth"], want["depth"])


@pytest.mark.parametrize("C,G,D", [(8, 4, 4), (16, 4, 4), (8, 8, 8), (16, 8, 4), (8, 4, 8), (16, 4, 8)])
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("regime", ["smooth", "random", "mixed", "planes"])
@needs_probes
def test_warp_agg_lds_window_variant_bit_identical(C, G, D, fuse, regime):
    """The LDS-staged source-window form (variant 5) against the one-thread form: equal bits whether every tap comes from
    the staged window (smooth hypotheses), almost none does (per-pixel random depths: the lane-by-lane buffer-load path),
    or a mixture -- incl. ragged tiles, views that leave the map, B = 2 and the saved softmax mass."""
    from mvster_amd.synthetic import make_inputs as mk
    for (h, w, nv, B) in ((37, 53, 3, 2), (64, 96, 4, 1), (5, 7, 2, 1), (40, 130, 2, 1)):
        g = torch.Generator().manual_seed(C * 131 + D * 7 + h)
        _, proj, dv = mk(nviews=nv + 1, H=h * 8, W=w * 8, batch=B, seed=h, rotate=True)
        pm = proj["stage4"].clone()
        # stage-4 intrinsics are for the full-resolution map of an (8h x 8w) image: rescale to this h x w map
        pm[:, :, 1, :2, :] = pm[:, :, 1, :2, :] / 8.0
        lo, hi = dv[:, :1, None, None], dv[:, -1:, None, None]
        yy = torch.linspace(0, 1, h).view(1, 1, h, 1)
        xx = torch.linspace(0, 1, w).view(1, 1, 1, w)
        plane = lo + (hi - lo) * (0.2 + 0.5 * xx + 0.2 * yy)                   # a slanted plane
        step = (hi - lo) / 64 * torch.arange(D).view(1, D, 1, 1)
        if regime == "smooth":
            hypo = (plane + step).expand(B, D, h, w).clone()
        elif regime == "random":
            hypo = lo + (hi - lo) * torch.rand(B, D, h, w, generator=g)
        elif regime == "mixed":
            hypo = (plane + step).expand(B, D, h, w).clone()
            out = torch.rand(B, 1, h, w, generator=g) < 0.1                   # 10 % outlier pixels
            hypo = torch.where(out, lo + (hi - lo) * torch.rand(B, D, h, w, generator=g), hypo)
        else:                                                                  # two planes with a depth step in the tile
            hypo = (plane + step + (hi - lo) * 0.3 * (xx > 0.47).float()).expand(B, D, h, w).clone()
        ref = torch.randn(B, h, w, C, generator=g)
        src = torch.randn(nv, B, h, w, C, generator=g)
        rt = ops.relative_projection(pm.to(DEV))
        args = (ref.to(DEV), src.to(DEV), rt, hypo.contiguous().to(DEV), G, True, fuse, 2.0)
        base, wbase = ops.warp_agg_fwd_cl(*args, want_wsum=True, variant=1)
        assert torch.isfinite(base).all() and base.abs().max() > 0
        o, ws = ops.warp_agg_fwd_cl(*args, want_wsum=True, variant=5)
        from mvster_amd import _lib
        assert _lib.last_kernel().startswith("warp_agg_fwd_tile_kernel<")
        assert torch.equal(o, base), (C, G, D, h, w, regime, (o - base).abs().max().item())
        assert torch.equal(ws, wbase), (C, G, D, h, w, regime)


@pytest.mark.parametrize("case", ["a", "b", "c", "d", "e"])
def test_warp_agg_vs_reference_homo_warping(golden, case):
    """The reference's homo_warping outputs (fixture G1) pushed through the rest of the stage arithmetic on the CPU, against
    the fused kernel: a = DTU-like cameras, b = source map smaller than the reference map, c = far-away camera (mostly
    zero padding), d = z == 0 exactly (-> 1e-9, mvs4net_utils.py:38-39), e = B=2 with per-sample cameras."""
    g = golden("g1_warp")
    fea = g.t({"a": "a_fea", "b": "b_fea", "c": "c_fea", "d": "c_fea", "e": "e_fea"}[case])     # [B,8,Hs,Ws]
    depth = g.t({"d": "d_depth", "e": "e_depth"}.get(case, "a_depth"))                          # [B,4,32,40]
    warped = g.t(case + "_out")                                                                 # [B,8,4,32,40]
    B = fea.shape[0]
    if case == "d":
        rt = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]).view(1, 1, 12)
    else:
        ref_p = g.t("e_ref" if case == "e" else "a_ref")
        src_p = g.t({"c": "c_src", "e": "e_src"}.get(case, "a_src"))
        P = torch.matmul(src_p, torch.inverse(ref_p))
        rt = torch.cat([P[:, :3, :3].reshape(-1, 9), P[:, :3, 3]], 1).unsqueeze(1).contiguous()
    torch.manual_seed(3)
    ref_fea = torch.randn(B, 8, 32, 40)
    cor = (warped.reshape(B, 4, 2, 4, 32, 40) * ref_fea.unsqueeze(2).repeat(1, 1, 4, 1, 1).reshape(B, 4, 2, 4, 32, 40)).mean(2)
    wgt = torch.softmax(cor.sum(1) / 2.0, 1) / np.sqrt(8)
    want = (wgt.unsqueeze(1) * cor) / (1e-8 + wgt).unsqueeze(1)
    base = None
    for variant in (1, 0):
        out = ops.warp_agg_fwd_cl(ops.to_channels_last(ref_fea.to(DEV)), ops.to_channels_last(fea.to(DEV)).unsqueeze(0),
                                  rt.to(DEV), depth.to(DEV), 4, True, True, 2.0, variant=variant)
        assert torch.isfinite(out).all()
        err = (out.permute(0, 4, 1, 2, 3).cpu() - want).abs().max().item()
        assert err <= 5e-6 * max(want.abs().max().item(), 1.0), (case, variant, err)
        base = out if base is None else base
    note("warp_agg_g1_" + case, max_abs=err)


CONV_CASES = [
    ("c3d_133_8_8", dict(cin=8, cout=8, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1)), (2, 8, 4, 20, 28)),
    ("c3d_133_4_8", dict(cin=4, cout=8, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1)), (1, 4, 4, 18, 22)),
    ("c3d_133s2_8_16", dict(cin=8, cout=16, k=(1, 3, 3), s=(1, 2, 2), p=(0, 1, 1)), (2, 8, 4, 20, 28)),
    ("c3d_333_16_16", dict(cin=16, cout=16, k=3, s=1, p=1), (1, 16, 4, 14, 18)),
    ("c3d_333_32_32", dict(cin=32, cout=32, k=3, s=1, p=1), (1, 32, 4, 10, 12)),
    ("c3d_333_64_64", dict(cin=64, cout=64, k=3, s=1, p=1), (1, 64, 8, 8, 10)),
    ("c3d_133s2_32_64", dict(cin=32, cout=64, k=(1, 3, 3), s=(1, 2, 2), p=(0, 1, 1)), (1, 32, 4, 12, 16)),
    ("c3d_333s2_16_32", dict(cin=16, cout=32, k=3, s=2, p=1), (1, 16, 8, 12, 16)),
    ("c2d_55s2_16_32", dict(cin=16, cout=32, k=(1, 5, 5), s=(1, 2, 2), p=(0, 2, 2)), (3, 16, 1, 44, 70)),
    ("c2d_33_64_8", dict(cin=64, cout=8, k=(1, 3, 3), s=(1, 1, 1), p=(0, 1, 1)), (2, 64, 1, 21, 45)),
]


@pytest.mark.parametrize("cin", [4, 8])
@pytest.mark.parametrize("shape", [(1, 1, 8, 32), (2, 3, 21, 45), (1, 4, 37, 130), (5, 1, 64, 96), (1, 2, 5, 7), (1, 8, 64, 80)])
def test_narrow_mfma_conv_against_fp64_and_valu(cin, shape):
    """Shift-packed MFMA form of the narrow 3x3 layers (variant 10, conv_narrow.hip) against an fp64 convolution and the VALU
    kernel (variant 3): both tile heights, one and two workgroups per CU, ReLU on / off, with and without the skip tensor,
    ragged tiles, maps narrower than a tile, several tiles per workgroup."""
    from mvster_amd import _lib
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin * 100 + H)
    w = torch.randn(8, cin, 1, 3, 3, generator=g) * 0.2
    x = torch.randn(B, D, H, W, cin, generator=g).to(DEV)
    skip = torch.randn(B, D, H, W, 8, generator=g).to(DEV)
    worst = 0.0
    for relu in (True, False):
        layer = cp.ConvLayer(w.to(DEV), False, (1, 1, 1), (0, 1, 1), bias=torch.randn(8, generator=g).to(DEV), relu=relu, cin_pad=cin)
        ref0 = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double().cpu(), w.double(), padding=(0, 1, 1))
        ref0 = ref0 * layer.scale[:8].double().cpu().view(1, 8, 1, 1, 1) + layer.shift[:8].double().cpu().view(1, 8, 1, 1, 1)
        ref0 = (ref0.clamp_min(0) if relu else ref0).permute(0, 2, 3, 4, 1)
        for sk in (None, skip):
            ref = ref0 if sk is None else ref0 + sk.double().cpu()
            sm = cp.SKIP_NONE if sk is None else cp.SKIP_ADD
            valu = layer(x, skip=sk, skip_mode=sm, tiles=(0, 0, 3))
            scale = ref.abs().max().item()
            for mt in (2, 4):
                for wpc in (1, 2):
                    got = layer(x, skip=sk, skip_mode=sm, tiles=(mt, wpc, 10))
                    assert _lib.last_kernel().startswith("conv_narrow_kernel<%d, %d, " % (cin, mt)), _lib.last_kernel()
                    e = (got.double().cpu() - ref).abs().max().item() / scale
                    ev = (got - valu).abs().max().item() / scale
                    worst = max(worst, e)
                    assert e < 2e-6 and ev < 2e-6, (cin, shape, relu, sk is not None, mt, wpc, e, ev)
    note("conv_narrow_mfma_%d_%s" % (cin, "x".join(map(str, shape))), err_over_max=worst)


@pytest.mark.parametrize("shape", [(1, 14, 64), (2, 21, 45), (1, 37, 130), (3, 5, 7), (5, 128, 160), (2, 100, 200), (1, 512, 640)])
def test_narrow_pair_conv_against_fp64_and_two_launches(shape, monkeypatch):
    """FPN conv0[0] -> conv0[1] in one launch (mvster_conv_narrow_pair: the 8-channel intermediate in LDS, zero outside the
    image = the second layer's padding) against the fp64 composition of the two layers and against the two launches of the
    shift-packed kernel: exact tiles, ragged tiles, maps smaller than one tile, several tiles per workgroup, ReLU on / off
    per layer, one and two workgroups per CU; and FpnPlan._conv0 dispatches it from NARROW_PAIR_MIN_PIXELS."""
    from mvster_amd import _lib
    NB, H, W = shape
    g = torch.Generator().manual_seed(7 * H + W)
    w1 = torch.randn(8, 3, 1, 3, 3, generator=g) * 0.3
    w2 = torch.randn(8, 8, 1, 3, 3, generator=g) * 0.2
    x = torch.randn(NB, 1, H, W, 4, generator=g)
    x[..., 3] = 0
    x = x.to(DEV)
    worst = 0.0
    for relu1, relu2 in ((True, True), (False, True), (True, False)):
        a = cp.ConvLayer(w1.to(DEV), False, (1, 1, 1), (0, 1, 1), bias=torch.randn(8, generator=g).to(DEV), relu=relu1, cin_pad=4)
        b = cp.ConvLayer(w2.to(DEV), False, (1, 1, 1), (0, 1, 1), bias=torch.randn(8, generator=g).to(DEV), relu=relu2)
        ref = x[..., :3].permute(0, 4, 1, 2, 3).double().cpu()
        for l, w in ((a, w1), (b, w2)):
            ref = torch.nn.functional.conv3d(ref, w.double(), padding=(0, 1, 1))
            ref = ref * l.scale[:8].double().cpu().view(1, 8, 1, 1, 1) + l.shift[:8].double().cpu().view(1, 8, 1, 1, 1)
            ref = ref.clamp_min(0) if l.relu else ref
        ref = ref.permute(0, 2, 3, 4, 1)
        two = b(a(x, tiles=(2, 0, 10)), tiles=(2, 0, 10))
        scale = ref.abs().max().item()
        for wpc in (1, 2):
            got = torch.full((NB, 1, H, W, 8), float("nan"), device=DEV)
            rc = _lib.load().mvster_conv_narrow_pair(x.data_ptr(), a.w_small.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr(),
                                                     b.w_small.data_ptr(), b.scale.data_ptr(), b.shift.data_ptr(), got.data_ptr(),
                                                     NB, H, W, int(relu1), int(relu2), wpc, ops._stream())
            _lib.check(rc, "conv_narrow_pair")
            assert _lib.last_kernel() == "conv_narrow_pair_kernel"
            e = (got.double().cpu() - ref).abs().max().item() / scale
            e2 = (got - two).abs().max().item() / scale
            worst = max(worst, e)
            assert e < 2e-6 and e2 < 2e-6, (shape, relu1, relu2, wpc, e, e2)
            # same MFMA chain per output as the two launches (the structural zeros of the shift packing are exact no-ops)
            assert torch.equal(got, two), (shape, relu1, relu2, wpc)
    note("conv_narrow_pair_%s" % "x".join(map(str, shape)), err_over_max=worst)

    plan = cp.FpnPlan.__new__(cp.FpnPlan)                        # the plan's dispatch: same result either way
    plan.conv0 = [a, b]
    fused = plan._conv0(x)
    fused_kernel = _lib.last_kernel()
    monkeypatch.setattr(cp, "FUSE_CONV0", False)
    plain = plan._conv0(x)
    assert (fused_kernel == "conv_narrow_pair_kernel") == (NB * H * W >= cp.NARROW_PAIR_MIN_PIXELS), fused_kernel
    assert _lib.last_kernel() != "conv_narrow_pair_kernel"
    assert (fused - plain).abs().max().item() <= 2e-6 * scale


@pytest.mark.parametrize("shape", [(1, 1, 8, 32), (2, 3, 21, 45), (1, 4, 37, 130), (1, 2, 5, 7), (2, 4, 128, 160)])
def test_narrow_mfma_conv_four_output_channels(shape):
    """The 8 -> 4 form of the shift-packed kernel (mvster_conv_narrow4: weights padded to eight output columns, the lanes of
    channels 4..7 store nothing, output pitch four) -- the input gradient of reg2d's first layer in training, i.e. the gradient
    of the cost volume's four groups: against an fp64 convolution and the direct kernel, both tile heights, ragged tiles; the
    plan takes it from 16 384 voxels on; a weight refreshed in place (the training step's re-pack) is followed."""
    from mvster_amd import _lib
    B, D, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    w = (torch.randn(4, 8, 1, 3, 3, generator=g) * 0.2).to(DEV)
    x = torch.randn(B, D, H, W, 8, generator=g).to(DEV)
    layer = cp.ConvLayer(w, False, (1, 1, 1), (0, 1, 1), relu=False)
    assert layer.w_small4 is not None and layer.w_small is None
    for rnd in range(2):
        ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), padding=(0, 1, 1)).permute(0, 2, 3, 4, 1)
        scale = ref.abs().max().item()
        direct = layer(x, tiles=(1, 1, 0))
        assert _lib.last_kernel().startswith("conv_mfma_kernel<")
        for mt in (2, 4):
            got = layer(x, tiles=(mt, 1, 10))
            assert _lib.last_kernel() == "conv_narrow_kernel<8, %d, %d, false, true>" % (mt, 4 if mt == 2 else 3), _lib.last_kernel()
            assert got.shape == direct.shape == (B, D, H, W, 4)
            assert (got.double() - ref).abs().max().item() < 2e-6 * scale
            assert (got - direct).abs().max().item() < 2e-6 * scale
        auto = layer(x)
        assert _lib.last_kernel().startswith("conv_narrow_kernel<" if B * D * H * W >= cp.NARROW_MIN_VOXELS else "conv_mfma_kernel<")
        assert (auto - direct).abs().max().item() < 2e-6 * scale
        w.mul_(-0.5).add_(0.01)                       # the optimizer's in-place update, then the step's re-pack
        layer.repack_on_device(w)


@pytest.mark.parametrize("cin", [4, 8])
def test_narrow_mfma_conv_non_finite_footprint(cin):
    """Pins the one documented divergence of the shift-packed MFMA kernel (variant 10, the default for narrow layers with
    >= 16 384 output voxels) on NON-FINITE inputs: its N operand multiplies structural-zero weights with real inputs (a pixel
    pair shares four tap columns), so a NaN / Inf input pixel at column c poisons, besides the reference's 3x3 footprint
    (rows y-1..y+1, columns c-1..c+1, which the VALU kernel reproduces exactly), ONE more output column of the same rows --
    c-2 for even c, c+2 for odd c (the other pixel of the pair on that side).  Nothing else changes, bit for bit.
    INTEGRATION.md ("Non-finite inputs") states this; finite inputs are unaffected (0 * finite = 0 exactly)."""
    B, D, H, W = 1, 1, 24, 96
    g = torch.Generator().manual_seed(7 + cin)
    w = torch.randn(8, cin, 1, 3, 3, generator=g) * 0.2
    layer = cp.ConvLayer(w.to(DEV), False, (1, 1, 1), (0, 1, 1), relu=False, cin_pad=cin)
    base = torch.randn(B, D, H, W, cin, generator=g).to(DEV)
    for (y, c, bad) in ((10, 40, float("nan")), (11, 41, float("inf")), (5, 33, float("nan")), (6, 62, float("-inf"))):
        x = base.clone()
        x[0, 0, y, c, 1] = bad
        x0 = base.clone()
        x0[0, 0, y, c, :] = 0.0
        for variant, extra in ((3, None), (10, c - 2 if c % 2 == 0 else c + 2)):
            got = layer(x, tiles=(0, 0, variant) if variant == 3 else (2, 1, 10))
            clean = layer(x0, tiles=(0, 0, variant) if variant == 3 else (2, 1, 10))
            nonfinite = ~torch.isfinite(got).all(-1)[0, 0]                    # [H, W]
            want = torch.zeros(H, W, dtype=torch.bool, device=DEV)
            want[y - 1:y + 2, c - 1:c + 2] = True
            allowed = want.clone()
            if extra is not None:
                allowed[y - 1:y + 2, extra] = True
            assert (nonfinite & want).sum() == want.sum(), (cin, variant, y, c)          # the reference's footprint
            stray = (nonfinite & ~allowed).nonzero().tolist()
            assert not stray, (cin, variant, y, c, stray[:8])
            if extra is not None:
                assert nonfinite[y - 1:y + 2, extra].all(), (cin, y, c, extra)        # the documented extra column IS there
            keep = ~allowed
            assert torch.equal(got[0, 0][keep], clean[0, 0][keep]), (cin, variant, y, c)


@needs_probes
@pytest.mark.parametrize("cin,cout,kd,shape", [(16, 16, 1, (2, 1, 37, 50)), (32, 32, 1, (1, 1, 64, 96)), (64, 64, 1, (1, 1, 9, 33)),
                                               (64, 32, 1, (2, 1, 21, 45)), (16, 16, 3, (1, 3, 21, 45)), (32, 32, 3, (1, 4, 32, 40)),
                                               (64, 64, 3, (1, 8, 8, 10)), (16, 16, 1, (5, 1, 128, 160))])
def test_bf16_split_conv_probe(cin, cout, kd, shape):
    """Probe (variant 11, conv_b3.hip): fp32 products as six bf16 MFMAs on 3-way split operands, fp32 accumulation.  Accuracy
    gate of the round-5 probe: against an fp64 convolution its error stays within the direct fp32 MFMA kernel's on the same
    layer (measured 0.44x), with and without the skip tensor, ragged tiles, one and three depth taps, both tile heights."""
    from mvster_amd import _lib
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin + 7 * cout + kd + H)
    w = torch.randn(cout, cin, kd, 3, 3, generator=g) * (2.0 / (cin * 9 * kd)) ** 0.5
    layer = cp.ConvLayer(w.to(DEV), False, (1, 1, 1), (kd // 2, 1, 1), bias=torch.randn(cout, generator=g).to(DEV) * 0.1, relu=True)
    x = torch.randn(B, D, H, W, cin, generator=g).to(DEV)
    skip = torch.randn(B, D, H, W, cout, generator=g).to(DEV)
    ref0 = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.to(DEV).double(), padding=(kd // 2, 1, 1))
    ref0 = ref0 * layer.scale[:cout].double().view(1, -1, 1, 1, 1) + layer.shift[:cout].double().view(1, -1, 1, 1, 1)
    ref0 = ref0.clamp_min(0).permute(0, 2, 3, 4, 1)
    worst = 0.0
    for sk in (None, skip):
        ref = ref0 if sk is None else ref0 + sk.double()
        sm = cp.SKIP_NONE if sk is None else cp.SKIP_ADD
        scale = ref.abs().max().item()
        e_dir = (layer(x, skip=sk, skip_mode=sm, tiles=(1, 1, 0)).double() - ref).abs().max().item() / scale
        for tyq in (1, 2):
            got = layer(x, skip=sk, skip_mode=sm, tiles=(tyq, 1, 11 | (1 << 8)))
            assert _lib.last_kernel().startswith("conv_b3_kernel<%d, " % cin), _lib.last_kernel()
            e = (got.double() - ref).abs().max().item() / scale
            worst = max(worst, e / max(e_dir, 1e-12))
            assert e <= max(1.0 * e_dir, 3e-7), (cin, cout, kd, shape, sk is not None, tyq, e, e_dir)
    note("conv_b3_probe_%d_%d_k%d_%s" % (cin, cout, kd, "x".join(map(str, shape))), err_over_direct_fp32=worst)


@pytest.mark.parametrize("C,G,D,h,w,B", [(64, 8, 8, 16, 24, 1), (32, 8, 8, 32, 48, 2), (16, 4, 4, 64, 96, 1), (8, 4, 4, 128, 192, 1),
                                         (8, 4, 4, 66, 50, 2)])
def test_warp_with_fused_hypothesis_scheduling_is_bit_identical(C, G, D, h, w, B):
    """mvster_warp_agg_fwd_sched: the stage's hypotheses computed inside the warp launch (schedule_inverse_range of the
    previous stage's inverse bounds, or init_inverse_range of depth_values at stage 1) against the two launches
    (models/mvs4net_utils.py:71-86 + :1025-1060): same hypotheses and same aggregated correlation, bit for bit."""
    g = torch.Generator().manual_seed(C + h)
    NV = 3
    ref = torch.randn(B, h, w, C, generator=g).to(DEV)
    src = torch.randn(NV, B, h, w, C, generator=g).to(DEV)
    _, proj, dv = make_inputs(nviews=NV + 1, H=8 * h, W=8 * w, seed=3, batch=B)
    rt = ops.relative_projection(proj["stage1"].to(DEV))
    dv = dv.to(DEV)
    # stage-1 form: init_inverse_range
    hypo = ops.init_range(dv, D, h, w, inverse=True)
    want = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0)
    got = ops.warp_agg_fwd_sched_cl(ref, src, rt, G, D, True, 2.0, depth_values=dv)
    assert got is not None and torch.equal(got[1], hypo) and torch.equal(got[0], want)
    # later stages: schedule_inverse_range of half-resolution bounds
    inv_max = (1.0 / (500 + 300 * torch.rand(B, h // 2, w // 2, generator=g))).to(DEV)
    inv_min = inv_max + (2e-4 * torch.rand(B, h // 2, w // 2, generator=g)).to(DEV)
    hypo = ops.schedule_inverse_range(inv_min, inv_max, D, h, w)
    want = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0)
    got = ops.warp_agg_fwd_sched_cl(ref, src, rt, G, D, True, 2.0, inv_min=inv_min, inv_max=inv_max)
    assert got is not None and torch.equal(got[1], hypo) and torch.equal(got[0], want)
    # a shape the wave-local kernel does not cover: the caller is told to run the two launches
    assert ops.warp_agg_fwd_sched_cl(ref[..., :8].contiguous(), src[..., :8].contiguous(), rt, 8, D, True, 2.0, depth_values=dv) is None


@pytest.mark.parametrize("name,cfg,shape", CONV_CASES)
def test_conv_bn_relu(name, cfg, shape):
    torch.manual_seed(hash(name) % 1000)
    m = M.ConvBnReLU3D(cfg["cin"], cfg["cout"], kernel_size=cfg["k"], stride=cfg["s"], pad=cfg["p"])
    m.load_state_dict(randomize_state(m.state_dict(), seed=3))
    m.eval()
    x = torch.randn(*shape)
    with torch.no_grad():
        want = cl5(twin(O._CBR3d(cfg["cin"], cfg["cout"], kernel_size=cfg["k"], stride=cfg["s"], pad=cfg["p"]), m)(x))
    layer = cp._cbr3d(m.to(DEV))
    worst = 0.0
    for mt in (1, 2, 4):
        for nt in (1, 2, 4):
            if layer.ntile_total % nt:
                continue
            got = layer(cl5(x).to(DEV), tiles=(mt, nt)).cpu()
            err = (got - want).abs().max().item()
            worst = max(worst, err)
            assert err <= 2e-5 * want.abs().max().item(), (name, mt, nt, err)
    # VALU kernel for the narrow layers (cin in {4, 8} -> 8 channels, 3x3)
    if layer.w_small is not None:
        got = layer(cl5(x).to(DEV), tiles=(0, 0, 3)).cpu()
        err = (got - want).abs().max().item()
        worst = max(worst, err)
        assert err <= 2e-5 * want.abs().max().item(), (name, "small", err)
        skip = torch.randn_like(want)
        got = layer(cl5(x).to(DEV), skip=skip.to(DEV), skip_mode=cp.SKIP_ADD, tiles=(0, 0, 3)).cpu()
        assert (got - (want + skip)).abs().max().item() <= 2e-5 * want.abs().max().item(), (name, "small+skip")
    # split-K variant (the 4 waves of a workgroup share the tiles and split the K steps)
    if cfg["cin"] >= 16:
        for mt, nt in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2)):
            if layer.ntile_total % nt:
                continue
            got = layer(cl5(x).to(DEV), tiles=(mt, nt, 2)).cpu()
            err = (got - want).abs().max().item()
            worst = max(worst, err)
            assert err <= 2e-5 * want.abs().max().item(), (name, "splitk", mt, nt, err)
    # LDS-staged variant (ordinary convs, cin % 16 == 0, kernel width 3 or 5)
    lds_ok = cfg["cin"] % 16 == 0 and layer.kernel[2] in (3, 5)
    if lds_ok:
        from mvster_amd.conv_plan import LDS_BUDGET
        for mt in (2, 4):
            patch = layer.kernel[0] * ((2 * mt - 1) * layer.stride[1] + layer.kernel[1]) * (31 * layer.stride[2] + layer.kernel[2]) * 64
            if patch > LDS_BUDGET:
                continue
            for nt in (1, 2, 4):
                if layer.ntile_total % nt:
                    continue
                got = layer(cl5(x).to(DEV), tiles=(mt, nt, 1)).cpu()
                err = (got - want).abs().max().item()
                worst = max(worst, err)
                assert err <= 2e-5 * want.abs().max().item(), (name, "lds", mt, nt, err)
    # persistent kernels (conv_pers.hip): LDS-DMA double-buffered patches (variant 5) / 1x1 with resident weights (6);
    # what a family does not cover reports "unsupported" and is skipped
    pers_tested = 0
    for variant, mts in ((5, (2,)), (6, (1, 2))):
        for mt in mts:
            for nt in (1, 2, 4):
                if layer.ntile_total % nt or (variant == 6 and nt > 1):
                    continue
                for wpc in (0, 1):
                    try:
                        got = layer(cl5(x).to(DEV), tiles=(mt, nt, variant | (wpc << 8))).cpu()
                    except RuntimeError as e:
                        assert "unsupported" in str(e), e
                        continue
                    pers_tested += 1
                    err = (got - want).abs().max().item()
                    worst = max(worst, err)
                    assert err <= 2e-5 * want.abs().max().item(), (name, "persistent", variant, mt, nt, err)
    note("conv_" + name, max_abs=worst, ref_absmax=want.abs().max().item(), lds_tested=float(lds_ok), persistent_tested=float(pers_tested))


PERS_CASES = [  # (cin, cout, kernel, stride, nt, input [B,D,H,W]): every instance of the persistent family, ragged sizes
    (16, 16, (1, 3, 3), (1, 1, 1), 1, (2, 1, 70, 100)), (16, 16, (1, 3, 3), (1, 1, 1), 1, (1, 1, 4, 33)),
    (16, 64, (1, 3, 3), (1, 1, 1), 1, (1, 1, 37, 65)),
    (32, 32, (1, 3, 3), (1, 1, 1), 2, (3, 1, 64, 64)), (32, 64, (1, 3, 3), (1, 1, 1), 2, (1, 1, 21, 50)),
    (64, 64, (1, 3, 3), (1, 1, 1), 1, (2, 1, 37, 50)), (64, 32, (1, 3, 3), (1, 1, 1), 1, (1, 1, 16, 33)),
    (32, 32, (1, 3, 3), (1, 1, 1), 1, (1, 1, 20, 40)),
    (16, 32, (1, 5, 5), (1, 2, 2), 2, (2, 1, 70, 100)), (16, 32, (1, 5, 5), (1, 2, 2), 2, (1, 1, 8, 34)),
    (16, 16, (3, 3, 3), (1, 1, 1), 1, (2, 4, 38, 70)), (16, 16, (3, 3, 3), (1, 1, 1), 1, (1, 3, 5, 31)),
    (16, 32, (1, 3, 3), (1, 2, 2), 2, (2, 4, 70, 100)), (16, 32, (1, 3, 3), (1, 2, 2), 2, (1, 1, 6, 66)),
    # 8 -> 16 channels, stride 2 (conv_pers8_kernel): two taps per 16-wide K step, odd sizes, a map smaller than a tile
    (8, 16, (1, 5, 5), (1, 2, 2), 1, (2, 1, 70, 100)), (8, 16, (1, 5, 5), (1, 2, 2), 1, (1, 1, 7, 33)),
    (8, 16, (1, 3, 3), (1, 2, 2), 1, (1, 4, 38, 70)), (8, 16, (1, 3, 3), (1, 2, 2), 1, (3, 1, 4, 6)),
    (64, 144, (1, 1, 1), (1, 1, 1), 1, (2, 1, 37, 53)), (64, 72, (1, 1, 1), (1, 1, 1), 1, (2, 1, 37, 53)),
    (32, 64, (1, 1, 1), (1, 1, 1), 1, (1, 2, 9, 17)), (64, 16, (1, 1, 1), (1, 1, 1), 1, (1, 1, 64, 80)),
]


@pytest.mark.parametrize("cin,cout,kernel,stride,nt,shape", PERS_CASES)
@pytest.mark.parametrize("with_skip", [False, True])
def test_persistent_conv_bit_identical_to_direct(cin, cout, kernel, stride, nt, shape, with_skip):
    """The persistent kernels accumulate in the packed K order (tap-major, channel-minor) like the direct kernel and share
    its epilogue arithmetic: outputs must be EQUAL, for every workgroups-per-CU setting (the grid changes which workgroup
    walks which tiles) and with the fused skip / ReLU."""
    g = torch.Generator().manual_seed(cin * 131 + cout + kernel[0] + shape[2])
    w = (torch.randn(cout, cin, *kernel, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, False, stride, tuple(k // 2 for k in kernel), relu=not with_skip)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    x = torch.randn(*shape, cin, generator=g).to(DEV)
    want = layer(x, tiles=(1, 1, 0))
    skip = torch.randn(want.shape, generator=g).to(DEV) if with_skip else None
    sm = cp.SKIP_ADD if with_skip else cp.SKIP_NONE
    if with_skip:
        want = layer(x, skip=skip, skip_mode=sm, tiles=(1, 1, 0))
    variant = 6 if kernel == (1, 1, 1) else 5
    for mt in ((1, 2) if variant == 6 else (2,)):
        # (32 + n: waves 4-7 of a 512-thread workgroup issue the LDS-DMA; built for the stride-2 families)
        for wpc in (0, 1, 2, 4) + ((33, 34) if variant == 5 and stride[2] == 2 else ()):
            got = layer(x, skip=skip, skip_mode=sm, tiles=(mt, nt, variant | (wpc << 8)))
            assert torch.equal(got, want), (mt, wpc, (got - want).abs().max().item())
    from mvster_amd import _lib
    assert _lib.last_kernel().startswith("conv1x1_pers_kernel<" if variant == 6 else "conv_pers8_kernel<" if cin == 8 else "conv_pers_kernel<")


@pytest.mark.parametrize("cin,cout,shape", [(64, 32, (2, 3, 9, 21)), (32, 16, (1, 1, 4, 33)), (32, 16, (1, 4, 16, 40)),
                                            (64, 64, (3, 1, 5, 64)), (32, 32, (1, 2, 7, 31))])
@pytest.mark.parametrize("with_skip", [False, True])
def test_persistent_transposed_conv_bit_identical_to_direct(cin, cout, shape, with_skip):
    """conv_tpers_kernel computes the four output-parity classes of a transposed 1x3x3 stride-(1,2,2) layer from one staged
    input tile, each in its packed K order: EQUAL to the direct kernel (one launch group per class), ragged sizes, with the
    U-Net skip added at the output resolution."""
    g = torch.Generator().manual_seed(cin + cout + shape[3])
    w = (torch.randn(cin, cout, 1, 3, 3, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, True, (1, 2, 2), (0, 1, 1), relu=True)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    x = torch.randn(*shape, cin, generator=g).to(DEV)
    want = layer(x, tiles=(1, 1, 0))
    skip = torch.randn(want.shape, generator=g).to(DEV) if with_skip else None
    sm = cp.SKIP_ADD if with_skip else cp.SKIP_NONE
    if with_skip:
        want = layer(x, skip=skip, skip_mode=sm, tiles=(1, 1, 0))
    for wpc in (0, 1, 2):
        got = layer(x, skip=skip, skip_mode=sm, tiles=(2, 1, 5 | (wpc << 8)))
        assert torch.equal(got, want), (wpc, (got - want).abs().max().item())
    from mvster_amd import _lib
    assert _lib.last_kernel().startswith("conv_tpers_kernel<")


@pytest.mark.parametrize("cin,cout,shape", [(16, 8, (1, 1, 7, 33)), (16, 8, (2, 1, 12, 40)), (32, 16, (3, 1, 5, 70)),
                                            (32, 16, (1, 2, 16, 32)), (16, 16, (2, 1, 9, 21)), (32, 32, (1, 1, 4, 64))])
@pytest.mark.parametrize("with_skip", [False, True])
def test_persistent_transposed_5x5_conv_bit_identical_to_direct(cin, cout, shape, with_skip):
    """conv_tpers_kernel<., ., 5>: the transposed 1x5x5 stride-(1,2,2) layers (the input gradients of the FPN's 5x5 stride-2
    convolutions; models/mvs4net_utils.py:430-446 in training) -- output-parity classes of 9 / 6 / 6 / 4 taps from one input
    tile staged with a one-pixel ring, each in its packed K order: EQUAL to the direct kernel, ragged sizes, 8 output channels
    (half an N tile), with and without a tensor added in the epilogue; and the plan picks it for maps of >= 40 960 voxels."""
    from mvster_amd import _lib
    g = torch.Generator().manual_seed(cin + cout + shape[3])
    w = (torch.randn(cin, cout, 1, 5, 5, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, True, (1, 2, 2), (0, 2, 2), relu=False)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    x = torch.randn(*shape, cin, generator=g).to(DEV)
    B, D, H, W = shape
    skip = torch.randn(B, D, 2 * H, 2 * W, cout, generator=g).to(DEV) if with_skip else None
    sm = cp.SKIP_ADD if with_skip else cp.SKIP_NONE
    want = layer(x, skip=skip, skip_mode=sm, tiles=(1, 1, 0))
    assert _lib.last_kernel().startswith("conv_mfma_kernel<")
    for wpc in (0, 1, 2):
        got = layer(x, skip=skip, skip_mode=sm, tiles=(2, 1, 5 | (wpc << 8)))
        assert torch.equal(got, want), (wpc, (got - want).abs().max().item())
    assert _lib.last_kernel().startswith("conv_tpers_kernel<") and _lib.last_kernel().endswith(", 5>")
    if (cin, cout) == (16, 8) and not with_skip:
        big = torch.randn(2, 1, 128, 160, cin, generator=g).to(DEV)
        got = layer(big)
        assert _lib.last_kernel().startswith("conv_tpers_kernel<"), _lib.last_kernel()
        assert torch.equal(got, layer(big, tiles=(1, 1, 0)))


@pytest.mark.parametrize("cin,cout,shape", [(32, 64, (2, 1, 18, 34)), (64, 64, (1, 1, 8, 66)), (32, 16, (3, 1, 4, 6)),
                                            (16, 64, (2, 1, 10, 38))])
def test_persistent_1x1_with_upsample_add_bit_identical_to_direct(cin, cout, shape):
    """FPN's lateral step -- inner(conv) + bilinear x2 (align_corners) up-sampling of the coarser level -- in the epilogue of
    the persistent 1x1 kernel: the direct kernel's arithmetic, so EQUAL."""
    g = torch.Generator().manual_seed(cin + cout)
    w = (torch.randn(cout, cin, 1, 1, 1, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, False, (1, 1, 1), (0, 0, 0), relu=False)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    x = torch.randn(*shape, cin, generator=g).to(DEV)
    B, D, H, W = shape
    prev = torch.randn(B, 1, H // 2, W // 2, cout, generator=g).to(DEV)
    want = layer(x, skip=prev, skip_mode=cp.SKIP_UPSAMPLE_ADD, tiles=(1, 1, 0))
    for mt in ((2, 4) if cin == 16 else (1, 2)):
        for wpc in (0, 1, 3):
            got = layer(x, skip=prev, skip_mode=cp.SKIP_UPSAMPLE_ADD, tiles=(mt, 1, 6 | (wpc << 8)))
            assert torch.equal(got, want), (mt, wpc, (got - want).abs().max().item())
    from mvster_amd import _lib
    assert _lib.last_kernel().startswith("conv1x1_pers_kernel<")


PP_CASES = [(16, 16, (1, 1, 1), 1, (2, 1, 70, 100)), (16, 16, (1, 1, 1), 1, (1, 1, 4, 33)), (32, 32, (1, 1, 1), 2, (3, 1, 64, 64)),
            (32, 32, (1, 1, 1), 2, (7, 1, 12, 64)), (16, 32, (1, 2, 2), 2, (2, 4, 70, 100)), (16, 32, (1, 2, 2), 2, (1, 1, 5, 200))]


@pytest.mark.parametrize("cin,cout,stride,nt,shape", PP_CASES)
@pytest.mark.parametrize("with_skip", [False, True])
@needs_probes
def test_pingpong_conv_bit_identical_to_direct(cin, cout, stride, nt, shape, with_skip):
    """Variant 7 (eight waves, the two halves of a workgroup in anti-phase) walks the same K order: EQUAL to the direct
    kernel, odd and even tile counts per workgroup, workgroups with no tile at all."""
    g = torch.Generator().manual_seed(cin * 17 + cout + shape[2])
    w = (torch.randn(cout, cin, 1, 3, 3, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, False, stride, (0, 1, 1), relu=not with_skip)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    x = torch.randn(*shape, cin, generator=g).to(DEV)
    want = layer(x, tiles=(1, 1, 0))
    skip = torch.randn(want.shape, generator=g).to(DEV) if with_skip else None
    sm = cp.SKIP_ADD if with_skip else cp.SKIP_NONE
    if with_skip:
        want = layer(x, skip=skip, skip_mode=sm, tiles=(1, 1, 0))
    got = layer(x, skip=skip, skip_mode=sm, tiles=(2, nt, 7))
    assert torch.equal(got, want), (got - want).abs().max().item()
    from mvster_amd import _lib
    assert _lib.last_kernel().startswith("conv_pp_kernel<")


WINO_CASES = [  # (cin, cout, kd, nt, variant, input [B,D,H,W]): every instance of the Winograd families, ragged sizes
    (16, 16, 1, 1, 8, (2, 1, 70, 100)), (16, 16, 1, 1, 8, (1, 1, 4, 33)), (16, 16, 1, 1, 8, (7, 1, 13, 63)),
    (16, 32, 1, 2, 8, (1, 1, 37, 65)), (32, 32, 1, 2, 8, (3, 1, 64, 64)), (32, 32, 1, 2, 8, (1, 4, 38, 70)),
    (32, 16, 1, 1, 8, (1, 1, 5, 200)), (32, 64, 1, 2, 8, (1, 1, 21, 50)),
    # ring form: depth taps (with the padded ones skipped), 64 channels, both roles of waves 4-7, dead steps at the end
    (16, 16, 3, 1, 9, (2, 4, 38, 70)), (16, 16, 3, 1, 9, (1, 1, 5, 31)), (16, 16, 3, 1, 9, (1, 2, 8, 32)),
    (32, 32, 3, 2, 9, (1, 8, 16, 20)), (32, 32, 3, 1, 9, (2, 3, 9, 40)), (64, 64, 3, 2, 9, (1, 4, 13, 63)),
    (64, 64, 3, 1, 9, (1, 8, 8, 10)), (64, 64, 1, 2, 9, (5, 1, 20, 40)), (64, 32, 1, 1, 9, (2, 1, 37, 50)),
    (64, 32, 1, 2, 9, (1, 1, 16, 33)), (16, 16, 1, 1, 9, (1, 1, 4, 33)), (32, 32, 1, 2, 9, (3, 1, 64, 64)),
    # ... mode 2 (variant word 9 | 1 << 8): two N tiles per compute wave, loading waves
    (32, 32, 1, 2, 265, (3, 1, 64, 64)), (64, 64, 3, 2, 265, (1, 4, 13, 63)), (64, 32, 1, 2, 265, (2, 1, 37, 50)),
    (16, 32, 1, 2, 265, (1, 1, 37, 65)), (32, 64, 3, 2, 265, (1, 8, 16, 20)), (64, 64, 1, 2, 265, (1, 1, 8, 32)),
]


@pytest.mark.parametrize("cin,cout,kd,nt,variant,shape", WINO_CASES)
@pytest.mark.parametrize("with_skip", [False, True])
def test_winograd_conv_against_fp64_and_direct(cin, cout, kd, nt, variant, shape, with_skip):
    """Variants 8 / 9 compute the 3x3 in-plane taps as F(2x2, 3x3) minimal filtering: a different operation order, so the
    comparison is against an fp64 convolution -- the error must stay within 2e-6 of max |y| (the direct kernel's own error on
    these inputs is ~4e-7) -- and within 2e-6 of the direct kernel."""
    g = torch.Generator().manual_seed(cin * 29 + cout + shape[3] + kd)
    w = (torch.randn(cout, cin, kd, 3, 3, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, False, (1, 1, 1), (kd // 2, 1, 1), relu=not with_skip)
    assert layer.wino_eligible() and layer.wpk_wino is not None
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    x = torch.randn(*shape, cin, generator=g).to(DEV)
    B, D, H, W = shape
    ref = F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), padding=(kd // 2, 1, 1))
    ref = ref * layer.scale[:cout].double().view(1, -1, 1, 1, 1) + layer.shift[:cout].double().view(1, -1, 1, 1, 1)
    if not with_skip:
        ref = ref.clamp_min(0)
    ref = ref.permute(0, 2, 3, 4, 1)
    skip = torch.randn(B, D, H, W, cout, generator=g).to(DEV) if with_skip else None
    sm = cp.SKIP_ADD if with_skip else cp.SKIP_NONE
    if with_skip:
        ref = ref + skip.double()
    direct = layer(x, skip=skip, skip_mode=sm, tiles=(1, 1, 0))
    scale = ref.abs().max().item()
    for wpc in ((0, 1, 2) if variant == 8 else (0,)):
        got = layer(x, skip=skip, skip_mode=sm, tiles=(2, nt, variant | (wpc << 8)))      # (265 = 9 with its mode bit set)
        assert torch.isfinite(got).all()
        assert (got.double() - ref).abs().max().item() <= 2e-6 * scale, (wpc, (got.double() - ref).abs().max().item() / scale)
        assert (got - direct).abs().max().item() <= 2e-6 * scale
    from mvster_amd import _lib
    assert _lib.last_kernel().startswith("conv_wino_kernel<" if variant == 8 else "conv_wino_ring_kernel<")
    if variant == 265:
        assert _lib.last_kernel().endswith(", 2>")
    note("conv_winograd%d_%d_%d_k%d_nt%d_%s" % (variant, cin, cout, kd, nt, "x".join(map(str, shape))), err_over_max=(got.double() - ref).abs().max().item() / scale,
         direct_err_over_max=(direct.double() - ref).abs().max().item() / scale)


@pytest.mark.parametrize("cin,shape", [(16, (1, 4, 38, 70)), (16, (2, 2, 9, 40)), (16, (1, 6, 8, 32)), (16, (3, 4, 13, 63)),
                                       (32, (1, 8, 16, 20)), (32, (1, 4, 38, 70)), (32, (2, 2, 8, 40)), (32, (1, 6, 8, 32))])
@pytest.mark.parametrize("with_skip", [False, True])
def test_winograd_pair_form_is_the_ring_kernel_bit_for_bit(cin, shape, with_skip):
    """conv_wino_pair_kernel (variant word 9 | 3 << 8, nt = 1): two output slices per tile, each input slice of the
    four-slice window fetched and transformed once for both.  Per output element the (depth tap, chunk) order is the ring
    kernel's, so the results are equal bit for bit -- ragged windows, first / last pairs (padded depth taps skipped), several
    tiles per workgroup, with and without the epilogue's skip.  (Measured no faster than the ring kernel: DESIGN.md 8.6; the
    plan does not pick it.)"""
    from mvster_amd import _lib
    g = torch.Generator().manual_seed(cin + shape[3])
    w = (torch.randn(cin, cin, 3, 3, 3, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, False, (1, 1, 1), (1, 1, 1), relu=not with_skip)
    layer.scale.copy_(torch.rand(layer.scale.shape, generator=g) + 0.5)
    layer.shift.copy_(torch.randn(layer.shift.shape, generator=g) * 0.1)
    x = torch.randn(*shape, cin, generator=g).to(DEV)
    skip = torch.randn(*shape, cin, generator=g).to(DEV) if with_skip else None
    sm = cp.SKIP_ADD if with_skip else cp.SKIP_NONE
    want = layer(x, skip=skip, skip_mode=sm, tiles=(2, 1, 9))
    assert _lib.last_kernel().startswith("conv_wino_ring_kernel<")
    got = layer(x, skip=skip, skip_mode=sm, tiles=(2, 1, 9 | (3 << 8)))
    assert _lib.last_kernel().startswith("conv_wino_pair_kernel<"), _lib.last_kernel()
    assert torch.equal(got, want), (got - want).abs().max().item()


def test_untuned_shapes_pick_winograd_when_the_map_is_large_enough():
    """A shape the measured table has never seen: the plan's rule sends eligible layers with enough 8 x 32 tiles to the
    Winograd kernels and leaves small maps on the direct / split-K kernels; either way the result is the direct kernel's
    within the Winograd bound."""
    from mvster_amd import _lib
    g = torch.Generator().manual_seed(11)
    for cin, cout, kd, shape, want in ((16, 16, 1, (7, 1, 72, 104), "conv_wino_kernel<"), (64, 64, 3, (1, 5, 72, 104), "conv_wino_ring_kernel<"),
                                       (32, 32, 3, (1, 3, 24, 40), None), (16, 16, 1, (1, 1, 24, 40), None)):
        w = (torch.randn(cout, cin, kd, 3, 3, generator=g) * 0.1).to(DEV)
        layer = cp.ConvLayer(w, False, (1, 1, 1), (kd // 2, 1, 1), relu=True)
        x = torch.randn(*shape, cin, generator=g).to(DEV)
        assert cp.layer_signature(layer, *shape, 0) not in cp._tuning()
        reach, cp.FAMILY_REACH = cp.FAMILY_REACH, -1.0      # (the rule under test is what is left when no family entry is near)
        try:
            got = layer(x)
        finally:
            cp.FAMILY_REACH = reach
        name = _lib.last_kernel()
        if want is not None:
            assert name.startswith(want), name
        else:
            assert not name.startswith("conv_wino"), name
        ref = layer(x, tiles=(1, 1, 0))
        # (two fp32 results of a K = 144 ... 1728 reduction in different orders: each is within ~1e-6 of max |y| of the exact
        #  value on maps of this size, see test_winograd_conv_against_fp64_and_direct)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err <= 4e-6, (cin, cout, kd, shape, name, err)


def test_winograd_weights_follow_in_place_repack():
    """The transformed weights are refreshed by the same call that refreshes the packed ones (training: once per step),
    also in the swapped / mirrored form of an input gradient."""
    g = torch.Generator().manual_seed(5)
    w = (torch.randn(16, 16, 1, 3, 3, generator=g) * 0.1).to(DEV)
    layer = cp.ConvLayer(w, False, (1, 1, 1), (0, 1, 1))
    x = torch.randn(1, 1, 9, 21, 16, generator=g).to(DEV)
    w2 = (torch.randn(16, 16, 1, 3, 3, generator=g) * 0.1).to(DEV)
    layer.repack_on_device(w2, swap=True, flip=True)
    want = F.conv2d(x[0].double().permute(0, 3, 1, 2), w2[:, :, 0].double().transpose(0, 1).flip(2, 3), padding=1).permute(0, 2, 3, 1)
    got = layer(x, tiles=(2, 1, 8))
    assert (got[0].double() - want).abs().max().item() <= 2e-6 * want.abs().max().item()


@pytest.mark.parametrize("cin,cout,k,pad,op,s", [(64, 32, (1, 3, 3), (0, 1, 1), (0, 1, 1), (1, 2, 2)),
                                                 (16, 8, (1, 3, 3), (0, 1, 1), (0, 1, 1), (1, 2, 2)),
                                                 (32, 16, (1, 3, 3), (0, 1, 1), (0, 1, 1), (1, 2, 2)),
                                                 (32, 16, 3, 1, 1, 2)])
def test_conv_transposed_skip(cin, cout, k, pad, op, s):
    torch.manual_seed(cin + cout)
    seq = M._deconv_bn_relu(cin, cout, k, pad, op, s)
    seq.load_state_dict(randomize_state(seq.state_dict(), seed=4))
    seq.eval()
    x = torch.randn(2, cin, 4, 9, 11)
    with torch.no_grad():
        y = twin(O._up3d(cin, cout, k, pad, op, s), seq)(x)
        skip = torch.randn_like(y)
        want = cl5(skip + y)
    layer = cp._up3d(seq.to(DEV))
    tile_sets = [(1, 1), (2, 1), (4, layer.ntile_total), (1, 1, 2), (2, 1, 2)]
    if layer.w_deconv is not None:
        tile_sets.append((0, 0, 4))          # VALU kernel
    for tiles in tile_sets:
        got = layer(cl5(x).to(DEV), skip=cl5(skip).to(DEV), skip_mode=cp.SKIP_ADD, tiles=tiles).cpu()
        err = (got - want).abs().max().item()
        assert err <= 2e-5 * want.abs().max().item(), (tiles, err)
    note("deconv_%d_%d" % (cin, cout), max_abs=err)


@pytest.mark.parametrize("name", ["reg2d_g8", "reg2d_g4", "reg3d_ds3", "reg3d_ds2"])
def test_reg_networks_vs_golden(golden, name):
    g = golden("g3_reg")
    sd = {k.split("/", 1)[1]: g.t(k) for k in g.keys() if k.startswith(name + "/")}
    x, want = g.t(name + "_x"), g.t(name + "_y")
    if name.startswith("reg2d"):
        m = M.reg2d(input_channel=x.shape[1], base_channel=8)
    else:
        m = M.reg3d(in_channels=x.shape[1], base_channels=8, down_size=int(name[-1]))
    m.load_state_dict(sd, strict=True)
    m.to(DEV).eval()
    xc = cl5(x).to(DEV)
    if name.startswith("reg2d"):
        plan = cp.Reg2dPlan(m, fuse_prob_into_conv11=False)
        logits = (plan(xc) @ plan.prob_w + plan.prob_b).cpu()
        fused = cp.Reg2dPlan(m)(xc).cpu()                  # prob head fused into conv11's epilogue
        assert (fused - want).abs().max().item() <= 5e-5 * want.abs().max().item()
    else:
        logits = cp.Reg3dPlan(m)(xc).cpu()
    err = (logits - want).abs().max().item()
    note("reg_" + name, max_abs=err, ref_absmax=want.abs().max().item())
    assert err <= 5e-5 * want.abs().max().item()


def test_fpn_plan_vs_oracle():
    torch.manual_seed(0)
    m = M.FPN4(base_channels=8)
    m.load_state_dict(randomize_state(m.state_dict(), seed=7))
    m.eval()
    img = torch.rand(3, 3, 64, 128)
    o = O.FPN4(base_channels=8)
    o.load_state_dict(m.state_dict(), strict=True)
    o.eval()
    with torch.no_grad():
        want = o(img)
    x = torch.zeros(3, 1, 64, 128, 4)
    x[:, 0, :, :, :3] = img.permute(0, 2, 3, 1)
    got = cp.FpnPlan(m.to(DEV))(x.to(DEV))
    for s in range(4):
        w = want["stage%d" % (s + 1)].permute(0, 2, 3, 1)
        err = (got[s][:, 0].cpu() - w).abs().max().item()
        note("fpn_stage%d" % (s + 1), max_abs=err, ref_absmax=w.abs().max().item())
        assert err <= 5e-5 * w.abs().max().item(), s


def test_fpn_tail_gather():
    from tests.conv_emulator import fpn_tail_gather_reference
    g = torch.Generator().manual_seed(2)
    for (NB, H, W, CO) in ((2, 12, 20, 8), (1, 64, 34, 8), (1, 8, 8, 16), (2, 40, 96, 8), (1, 18, 70, 8),
                           (2, 40, 96, 16), (1, 18, 70, 16), (3, 16, 64, 16), (1, 64, 132, 16)):     # (16 channels: 16 x 32 tiles)
        G = torch.randn(NB, 1, H // 2, W // 2, 9 * CO, generator=g)
        vb = torch.randn(9, CO, generator=g)
        want = fpn_tail_gather_reference(G, vb, H, W)
        got = ops.fpn_tail_gather(G.to(DEV), vb.to(DEV), H, W, separable=False).cpu()
        got2 = ops.fpn_tail_gather(G.to(DEV), vb.to(DEV), H, W, separable=True).cpu()
        err = max((got - want).abs().max().item(), (got2 - want).abs().max().item())
        note("fpn_tail_gather_%dx%d" % (H, W), max_abs=err, ref_absmax=want.abs().max().item())
        assert err <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("NB,H,W,CO,pitch", [(2, 12, 20, 8, 80), (1, 64, 34, 8, 72), (1, 8, 8, 16, 144), (1, 18, 70, 8, 80),
                                             (2, 32, 128, 8, 80), (1, 44, 132, 8, 80), (2, 70, 200, 8, 72)])   # (LDS-tiled form)
def test_fpn_tail_gather_adjoint(NB, H, W, CO, pitch):
    """mvster_fpn_tail_gather_bwd against autograd through the PyTorch restatement of the gather."""
    from tests.conv_emulator import fpn_tail_gather_reference
    g = torch.Generator().manual_seed(4)
    G = torch.randn(NB, 1, H // 2, W // 2, 9 * CO, generator=g, dtype=torch.float64, requires_grad=True)
    vb = torch.randn(9, CO, generator=g, dtype=torch.float64)
    gP = torch.randn(NB, 1, H, W, CO, generator=g)
    fpn_tail_gather_reference(G, vb, H, W).backward(gP.double())
    got = ops.fpn_tail_gather_bwd(gP.to(DEV), pitch=pitch).cpu()
    assert tuple(got.shape) == (NB, 1, H // 2, W // 2, pitch)
    err = (got[..., :9 * CO].double() - G.grad).abs().max().item()
    note("fpn_tail_gather_bwd_%dx%d" % (H, W), max_abs=err, ref_absmax=G.grad.abs().max().item())
    assert err <= 1e-5 * G.grad.abs().max().item()
    assert (got[..., 9 * CO:] == 0).all()


def test_fpn_lateral_up():
    from tests.conv_emulator import fpn_lateral_up_reference
    g = torch.Generator().manual_seed(3)
    for (NB, H, W, CI, CO) in ((2, 12, 20, 16, 72), (1, 64, 34, 16, 72), (2, 6, 130, 8, 72)):
        x = torch.randn(NB, 1, H, W, CI, generator=g)
        A = torch.randn(CO, CI, generator=g)
        bias = torch.randn(CO, generator=g)
        q = torch.randn(NB, 1, H // 2, W // 2, CO, generator=g)
        want = fpn_lateral_up_reference(x, A, bias, q)
        got = ops.fpn_lateral_up(x.to(DEV), A.to(DEV), bias.to(DEV), q.to(DEV)).cpu()
        err = (got - want).abs().max().item()
        note("fpn_lateral_up_%dx%d" % (H, W), max_abs=err, ref_absmax=want.abs().max().item())
        assert err <= 1e-5 * want.abs().max().item()


@pytest.mark.parametrize("NB,H,W", [(1, 16, 64), (2, 24, 100), (1, 40, 132), (5, 64, 96), (1, 128, 192), (1, 20, 68)])
def test_fpn_tail_fused_equals_the_two_launches(NB, H, W):
    """Finest FPN level: lateral step + gather-sum in one launch (the 72-channel map in LDS, lateral product on MFMA tiles)
    against mvster_fpn_lateral_up followed by mvster_fpn_tail_gather, and against the fp64 restatement of both: ragged tiles,
    maps barely inside the kernel's domain, several views."""
    from mvster_amd import _lib
    from tests.conv_emulator import fpn_lateral_up_reference, fpn_tail_gather_reference
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(NB, 1, H // 2, W // 2, 16, generator=g)
    A = torch.randn(72, 16, generator=g) * 0.3
    bias = torch.randn(72, generator=g)
    q = torch.randn(NB, 1, H // 4, W // 4, 72, generator=g)
    vb = torch.randn(9, 8, generator=g)
    two = ops.fpn_tail_gather(ops.fpn_lateral_up(x.to(DEV), A.to(DEV), bias.to(DEV), q.to(DEV)), vb.to(DEV), H, W)
    one = ops.fpn_tail_fused(x.to(DEV), A.to(DEV), bias.to(DEV), q.to(DEV), vb.to(DEV), H, W)
    assert one is not None and _lib.last_kernel() in ("fpn_tail_fused_kernel<16, %d>" % t for t in (16, 8, 32))     # (8, 32: probe switch)
    ref = fpn_tail_gather_reference(fpn_lateral_up_reference(x.double(), A.double(), bias.double(), q.double()), vb.double(), H, W)
    scale = ref.abs().max().item()
    e1 = (one.cpu().double() - ref).abs().max().item() / scale
    e2 = (two.cpu().double() - ref).abs().max().item() / scale
    d = (one - two).abs().max().item() / scale
    note("fpn_tail_fused_%dx%dx%d" % (NB, H, W), fused_vs_fp64=e1, two_launches_vs_fp64=e2, fused_vs_two=d)
    assert e1 <= 5e-6 and d <= 1e-6, (e1, e2, d)        # (measured: 1e-6 .. 2.2e-6 for both forms against fp64, 2-3e-7 between them)
    assert ops.fpn_tail_fused(x[:, :, :4, :30].contiguous().to(DEV), A.to(DEV), bias.to(DEV), q[:, :, :2, :15].contiguous().to(DEV),
                              vb.to(DEV), 8, 60) is None                 # outside the domain: the caller takes the two launches


BWD_CASES = [
    # C, G, D, group_cor, attn_fuse_d
    (64, 8, 8, True, True), (32, 8, 8, True, True), (16, 4, 4, True, True), (8, 4, 4, True, True),       # the shipped stages
    (8, 8, 8, True, True), (16, 8, 4, True, True), (32, 4, 8, True, True), (64, 4, 4, True, True),       # other group widths
    (8, 8, 4, False, True), (16, 16, 4, False, True), (32, 32, 4, False, True), (64, 64, 3, False, True),  # squared differences
    (16, 4, 4, True, False), (8, 4, 4, True, False), (8, 8, 5, False, False),                            # attn_fuse_d = False
    (8, 4, 16, True, True), (16, 4, 12, True, False),                                                    # up to 16 hypotheses
]


@pytest.mark.parametrize("C,G,D,group_cor,fuse", BWD_CASES)
def test_warp_agg_backward_vs_autograd(C, G, D, group_cor, fuse):
    """HIP backward (both kernels: row blocks with a scatter window, 64x4 tiles at C=8) vs PyTorch autograd through the
    CPU oracle's aggregate_views: B=2, a source map of another size than the reference map, a ragged pixel count, one
    view rolled in-plane by half a radian so that its taps leave the scatter window (the direct-atomic path)."""
    torch.manual_seed(C * 7 + D)
    B, N, h, w, Hs, Ws = 2, 3, 12, 70, 10, 60
    _, proj, dv = make_inputs(N, h * 8, w * 8, seed=9, batch=B)
    pm = proj["stage1"].clone()
    c, s_ = float(np.cos(0.5)), float(np.sin(0.5))
    roll = torch.tensor([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    pm[:, 2, 0, :3, :3] = roll @ pm[:, 2, 0, :3, :3]
    feats = [torch.randn(B, C, h, w, requires_grad=True)] + [torch.randn(B, C, Hs, Ws, requires_grad=True) for _ in range(N - 1)]
    hypo = O.init_inverse_range(dv, D, h, w) * (1 + 0.02 * torch.rand(B, D, h, w))
    cor = O.aggregate_views(feats, pm, hypo, group_cor, G, attn_temp=2.0, attn_fuse_d=fuse)
    gout = torch.randn_like(cor)
    cor.backward(gout)
    ref_cl = feats[0].detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    src_cl = torch.stack([f.detach() for f in feats[1:]]).permute(0, 1, 3, 4, 2).contiguous().to(DEV)
    rt = _oracle_rt(pm).to(DEV)
    Gk = G if group_cor else C
    out, wsum = ops.warp_agg_fwd_cl(ref_cl, src_cl, rt, hypo.to(DEV), Gk, group_cor, fuse, 2.0, want_wsum=True)
    e_fwd = (out.permute(0, 4, 1, 2, 3).cpu() - cor.detach()).abs().max().item() / max(cor.abs().max().item(), 1.0)
    scale = max(f.grad.abs().max().item() for f in feats)
    assert e_fwd <= 1e-4                        # (the forward has its own tests; measured 3e-6 .. 5e-5 here)
    # sorted scatter (the default: samples counting-sorted by source tile, no global atomics; written into a buffer full of
    # garbage: every texel is stored, none accumulated) / dense windows + gather pass / windows flushed with global atomics
    for form in ("sorted", "gather", "atomic"):
        into = None
        if form == "sorted":
            into = (torch.full_like(ref_cl, float("nan")), torch.full_like(src_cl, float("nan")))
        g_ref, g_src = ops.warp_agg_bwd_cl(ref_cl, src_cl, rt, hypo.to(DEV), out, wsum,
                                           gout.permute(0, 2, 3, 4, 1).contiguous().to(DEV), Gk, group_cor, fuse, 2.0,
                                           deterministic=form == "gather", sorted_scatter=form == "sorted", into=into)
        e_ref = (g_ref.permute(0, 3, 1, 2).cpu() - feats[0].grad).abs().max().item() / scale
        e_src = max((g_src[v].permute(0, 3, 1, 2).cpu() - feats[v + 1].grad).abs().max().item() for v in range(N - 1)) / scale
        note("warp_agg_bwd_C%d_G%d_D%d_%s_%s_%s" % (C, G, D, "group" if group_cor else "sqdiff", "fuse" if fuse else "nofuse", form),
             fwd_rel=e_fwd, ref_rel=e_ref, src_rel=e_src, grad_absmax=scale)
        assert e_ref <= 1e-4 and e_src <= 1e-4, form


@pytest.mark.parametrize("C,G,D,h,w", [(8, 4, 4, 64, 160), (16, 4, 4, 32, 80), (32, 8, 8, 16, 70)])
def test_warp_agg_backward_is_reproducible(C, G, D, h, w):
    """Inside a workgroup the gradients accumulate in integer (fixed-point) LDS counters, so the reference gradient is the
    same bits on every run; with ``deterministic=True`` the scatter windows are summed by the gather pass in fixed order
    and the source gradient is reproducible too, as long as every tap falls inside its workgroup's window (translated
    cameras here).  The default (windows flushed with global fp32 atomics) agrees to rounding."""
    torch.manual_seed(C + h)
    B, N = 2, 4
    _, proj, dv = make_inputs(N, h * 8, w * 8, seed=3, batch=B, rotate=False)
    rt = ops.relative_projection(proj["stage1"].to(DEV))
    ref = torch.randn(B, h, w, C, device=DEV)
    src = torch.randn(N - 1, B, h, w, C, device=DEV)
    hypo = (O.init_inverse_range(dv, D, h, w)[:, :1] * (1 + 0.002 * torch.arange(D).view(1, D, 1, 1)
                                                          + 0.0005 * torch.rand(B, D, h, w))).contiguous().to(DEV)
    out, wsum = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, want_wsum=True)
    gout = torch.randn_like(out)
    a = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, deterministic=True, sorted_scatter=False)
    b = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, deterministic=True, sorted_scatter=False)
    assert torch.equal(a[0], b[0]), "reference gradient"
    assert torch.equal(a[1], b[1]), "source gradient (gather pass)"
    c = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, deterministic=False, sorted_scatter=False)
    assert torch.equal(a[0], c[0])
    scale = c[1].abs().max().item()
    err = (a[1] - c[1]).abs().max().item() / scale
    note("warp_agg_bwd_gather_vs_atomic_C%d" % C, rel=err)
    assert err <= 1e-5
    # the sorted scatter: integer accumulation per source tile -- the same bits on every run whatever order the records land in
    s1 = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, sorted_scatter=True)
    s2 = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, sorted_scatter=True)
    assert torch.equal(s1[0], s2[0]) and torch.equal(s1[1], s2[1]), "sorted scatter"
    assert torch.equal(s1[0], a[0])
    err = (s1[1] - a[1]).abs().max().item() / scale
    note("warp_agg_bwd_sorted_vs_gather_C%d" % C, rel=err)
    assert err <= 1e-5


@pytest.mark.parametrize("C,G,D,h,w", [(8, 4, 4, 64, 160), (64, 8, 8, 16, 40)])
def test_warp_agg_backward_sorted_scatter_random_winners(C, G, D, h, w):
    """The regime the sorted scatter is for: every pixel's hypotheses anywhere in the depth range, unrelated to its
    neighbours' (what a cascade with random weights hands from stage to stage), rotated cameras.  Reproducible to the bit, and
    equal to the window / atomic form to rounding."""
    torch.manual_seed(C + w)
    B, N = 2, 4
    _, proj, dv = make_inputs(N, h * 8, w * 8, seed=5, batch=B)
    rt = ops.relative_projection(proj["stage1"].to(DEV))
    ref = torch.randn(B, h, w, C, device=DEV)
    src = torch.randn(N - 1, B, h, w, C, device=DEV)
    full = O.init_inverse_range(dv, 48, h, w)
    pick = torch.randint(0, 48 - D, (B, 1, h, w))
    hypo = torch.gather(full, 1, pick + torch.arange(D).view(1, D, 1, 1)).contiguous().to(DEV)
    out, wsum = ops.warp_agg_fwd_cl(ref, src, rt, hypo, G, True, True, 2.0, want_wsum=True)
    gout = torch.randn_like(out)
    s1 = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, sorted_scatter=True)
    s2 = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, sorted_scatter=True)
    assert torch.equal(s1[0], s2[0]) and torch.equal(s1[1], s2[1])
    c = ops.warp_agg_bwd_cl(ref, src, rt, hypo, out, wsum, gout, G, True, True, 2.0, sorted_scatter=False)
    assert torch.equal(s1[0], c[0])
    scale = c[1].abs().max().item()
    err = (s1[1] - c[1]).abs().max().item() / scale
    note("warp_agg_bwd_sorted_vs_atomic_random_C%d" % C, rel=err)
    assert err <= 1e-5


def test_cpu_tensor_is_rejected():
    with pytest.raises(RuntimeError):
        ops.relative_projection(torch.zeros(1, 2, 2, 4, 4))
