"""MVS4net on the GPU (all-HIP eval path, autograd train path) against the golden vectors
captured from the reference, the CPU oracle, and size-independent properties at full size."""
import json
import os

import numpy as np
import pytest
import torch

from mvster_amd import MVS4net, MVS4net_loss
from mvster_amd.graph import GraphedForward
from mvster_amd.synthetic import make_inputs
from oracle import mvs4_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPORT = {}


def note(name, **kv):
    REPORT[name] = {k: float(v) for k, v in kv.items()}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_model.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def to_dev(imgs, proj, dv):
    return [i.to(DEV) for i in imgs], {k: v.to(DEV) for k, v in proj.items()}, dv.to(DEV)


@pytest.fixture(scope="module")
def model(shipped_cfg, checkpoint):
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    return m.to(DEV).eval()


def test_eval_golden_teacher_forced(model, golden):
    """Every stage fed the reference's own hypotheses: continuous tensors tight, depth tie-aware."""
    g = golden("g6_eval")
    H, W, N = int(g.np("H")), int(g.np("W")), int(g.np("N"))
    imgs, proj, dv = to_dev(*make_inputs(nviews=N, H=H, W=W, seed=1))
    teacher = {"stage%d" % s: g.t("stage%d_hypo_depth" % s, DEV) for s in range(1, 5)}
    cap = {}
    out = model._forward_eval(imgs, proj, dv, teacher=teacher, capture=cap)
    for s in range(1, 5):
        name = "stage%d" % s
        st = out[name]
        cor = cap[name]["cor_feats"].cpu()
        want_cor = g.t("stage%d_cor_feats" % s)
        logits, want_logits = cap[name]["logits"].cpu(), g.t("stage%d_logits" % s)
        attn, want_attn = st["attn_weight"].cpu(), g.t("stage%d_attn_weight" % s)
        margin = g.t("stage%d_margin" % s)
        clear = margin > 1e-3
        depth, want_depth = st["depth"].cpu(), g.t("stage%d_depth" % s)
        flips = (depth != want_depth)
        note("teacher_" + name, cor_max=(cor - want_cor).abs().max(), cor_mean=(cor - want_cor).abs().mean(),
             cor_ref_absmax=want_cor.abs().max(), logits_max=(logits - want_logits).abs().max(),
             attn_max=(attn - want_attn).abs().max(), depth_l1_clear=(depth - want_depth)[clear].abs().mean(),
             depth_max_clear=(depth - want_depth)[clear].abs().max(), flip_rate=flips.float().mean(),
             flip_rate_clear=flips[clear].float().mean(), clear_frac=clear.float().mean())
        # measured on MI355X (profiles/r01_k_parity_model.json): max 1.4e-5 / mean 8e-8 of the scale, attn 5e-5
        assert (cor - want_cor).abs().max() <= 1e-4 * want_cor.abs().max()
        assert (cor - want_cor).abs().mean() <= 2e-6 * want_cor.abs().max()
        assert (attn - want_attn).abs().max() <= 3e-4
        assert (depth - want_depth)[clear].abs().mean() < 1e-4          # north-star tolerance: depth L1 < 1e-4
        conf, want_conf = st["photometric_confidence"].cpu(), g.t("stage%d_photometric_confidence" % s)
        assert conf.shape == want_conf.shape and (conf - want_conf).abs().max() <= 1e-3
        mono = st["mono_feat"].cpu()
        assert mono.shape == g.t("stage%d_mono_feat" % s).shape
        assert (mono - g.t("stage%d_mono_feat" % s)).abs().max() <= 5e-5 * g.t("stage%d_mono_feat" % s).abs().max()


def test_eval_golden_free_running(model, golden):
    """The cascade on its own hypotheses (API call, as test_mvs4.py:205 does): stage 1 is still
    teacher-free-identical; later stages are reported with the flip rate."""
    g = golden("g6_eval")
    H, W, N = int(g.np("H")), int(g.np("W")), int(g.np("N"))
    imgs, proj, dv = to_dev(*make_inputs(nviews=N, H=H, W=W, seed=1))
    out = model(imgs, proj, dv)
    assert set(out.keys()) >= {"stage1", "stage2", "stage3", "stage4", "depth", "photometric_confidence", "hypo_depth",
                               "attn_weight", "inverse_min_depth", "inverse_max_depth", "mono_feat"}
    assert torch.equal(out["depth"], out["stage4"]["depth"])
    for s in range(1, 5):
        st = out["stage%d" % s]
        want_depth = g.t("stage%d_depth" % s)
        depth = st["depth"].cpu()
        assert depth.shape == want_depth.shape
        close = (depth - want_depth).abs() <= 1e-3
        note("free_stage%d" % s, depth_l1=(depth - want_depth).abs().mean(), agree_frac=close.float().mean(),
             hypo_max=(st["hypo_depth"].cpu() - g.t("stage%d_hypo_depth" % s)).abs().max())
        assert st["photometric_confidence"].shape[-2:] == (H, W)
        # the free-running figure over ALL pixels, bounded (measured on G6: L1 0 / 1.2e-5 / 2.5e-5 / 1.15e-4 at stages 1-4,
        # 99.996 % of the stage-4 pixels within 1e-3).  The north star's "depth L1 < 1e-4" holds under the tie-aware
        # protocol only (clear-margin pixels, teacher forcing: test_eval_golden_teacher_forced, L1 = 0); all-pixel free
        # running at the last stage sits just above it because of winner-take-all flips on near-ties
        assert (depth - want_depth).abs().mean().item() < (1e-4 if s < 4 else 5e-4), s
        assert close.float().mean().item() > 0.9995, s
    # stage 1 hypotheses are input-independent of earlier stages: exact
    assert torch.equal(out["stage1"]["hypo_depth"].cpu(), g.t("stage1_hypo_depth"))
    s1 = out["stage1"]["depth"].cpu()
    clear = g.t("stage1_margin") > 1e-3
    assert (s1 - g.t("stage1_depth"))[clear].abs().mean() < 1e-4


@pytest.mark.parametrize("H,W,N", [(832, 1152, 5), (768, 1024, 3)])
def test_untuned_resolution_close_to_exhaustive_choice(model, H, W, N):
    """Resolutions the measured table does not hold (832 x 1152 x 5 is the reference's real "mid" workload, test_mvs4.py:41-42):
    the plans' own choices -- exact entries, else the layer family's entry nearest in size, else the heuristics -- summed over
    the distinct layers of a forward, against the best candidate per layer found by timing all of them here."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scripts.conv_microbench import auto_vs_best, record_eval_calls
    rows = auto_vs_best(record_eval_calls(model, H, W, N, torch.device(DEV)), torch.device(DEV), n=5)
    sa, sb = sum(r[1] for r in rows), sum(r[2] for r in rows)
    worst = max(rows, key=lambda r: r[1] / r[2])
    note("untuned_%dx%dx%d" % (H, W, N), auto_us=sa, best_us=sb, excess=sa / sb - 1, layers=len(rows),
         by_family=sum(1 for r in rows if r[4] == "family"), by_heuristic=sum(1 for r in rows if r[4] == "heuristic"),
         worst_excess=worst[1] / worst[2] - 1)
    print("untuned %dx%dx%d: worst layer %s" % (H, W, N, worst[0]))
    assert sa <= 1.05 * sb, (sa, sb, worst)


def test_graph_replay_is_bit_identical(model):
    imgs, proj, dv = to_dev(*make_inputs(nviews=5, H=128, W=192, seed=4))
    eager = model(imgs, proj, dv)
    eager = {k: v.clone() for k, v in eager.items() if torch.is_tensor(v)}
    gf = GraphedForward(model, imgs, proj, dv)
    out = gf()
    torch.cuda.synchronize()
    for k in ("depth", "attn_weight", "photometric_confidence", "hypo_depth"):
        assert torch.equal(out[k], eager[k]), k
    # new inputs through the static buffers
    imgs2, proj2, dv2 = to_dev(*make_inputs(nviews=5, H=128, W=192, seed=5))
    want = {k: v.clone() for k, v in model(imgs2, proj2, dv2).items() if torch.is_tensor(v)}
    out = gf(imgs2, proj2, dv2)
    torch.cuda.synchronize()
    assert torch.equal(out["depth"], want["depth"])


def test_packed_graph_inputs_one_copy_per_sample(model):
    """GraphedForward(packed=True): the static inputs are views of one flat device buffer in the layout of graph.pack_sample,
    so a new sample arrives with ONE host -> device copy (load_packed) -- same outputs as the eager forward on that sample."""
    from mvster_amd.graph import pack_sample
    a = make_inputs(nviews=4, H=128, W=192, seed=11)
    b = make_inputs(nviews=4, H=128, W=192, seed=12)
    gf = GraphedForward(model, *to_dev(*a), packed=True)
    assert gf.flat is not None and all(i.data_ptr() >= gf.flat.data_ptr() for i in gf.imgs)
    want_a = {k: v.clone() for k, v in model(*to_dev(*a)).items() if torch.is_tensor(v)}
    out = gf()
    torch.cuda.synchronize()
    assert torch.equal(out["depth"], want_a["depth"]) and torch.equal(out["attn_weight"], want_a["attn_weight"])
    host = pack_sample(*b)                                   # pinned, same layout
    assert host.is_pinned() and host.numel() == gf.flat.numel()
    gf.load_packed(host)
    out = gf()
    torch.cuda.synchronize()
    want_b = model(*to_dev(*b))
    assert torch.equal(out["depth"], want_b["depth"]) and torch.equal(out["photometric_confidence"], want_b["photometric_confidence"])
    with pytest.raises(RuntimeError):
        GraphedForward(model, *to_dev(*a)).load_packed(host)


FULL_SIZE = [(512, 640, 5),        # BASELINE config 2: DTU mid
             (1152, 1600, 5),      # config 3: DTU raw (1200x1600 cropped to a multiple of 64)
             (1024, 1920, 7)]      # config 5: Tanks & Temples (1920x1056 -> 1024 rows), 7 views


@pytest.mark.parametrize("H,W,N", FULL_SIZE)
def test_full_size_properties(model, H, W, N):
    """BASELINE configs 2, 3 and 5 at full size: properties that do not need the (slow) CPU oracle."""
    imgs, proj, dv = to_dev(*make_inputs(nviews=N, H=H, W=W, seed=0))
    out = model(imgs, proj, dv)
    out = {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()) for k, v in out.items()}
    out2 = model(imgs, proj, dv)
    for s in range(1, 5):
        st = out["stage%d" % s]
        D = st["hypo_depth"].shape[1]
        attn = st["attn_weight"]
        assert torch.isfinite(attn).all() and torch.isfinite(st["depth"]).all()
        assert (attn.sum(1) - 1).abs().max() <= 1e-5                       # softmax
        conf = st["photometric_confidence"]
        assert conf.shape == (1, H, W) and conf.min() >= 1.0 / D - 1e-6 and conf.max() <= 1 + 1e-6
        # winner-take-all: the depth is one of the hypotheses, the one with the largest weight
        idx = attn.max(1, keepdim=True)[1]
        assert torch.equal(torch.gather(st["hypo_depth"], 1, idx).squeeze(1), st["depth"])
        # idempotence / determinism of the whole eval path (no atomics in it)
        assert torch.equal(st["depth"], out2["stage%d" % s]["depth"])
        assert torch.equal(attn, out2["stage%d" % s]["attn_weight"])
        # inverse-depth bounds bracket the selected depth
        assert (st["inverse_min_depth"] >= 1 / st["depth"]).all() and (st["inverse_max_depth"] <= 1 / st["depth"]).all()
    # stage-1 hypotheses: index 0 is the farthest plane, exact endpoints
    h1 = out["stage1"]["hypo_depth"]
    assert (h1[:, 0] > h1[:, -1]).all()
    assert (h1[:, 0] - dv[0, 1]).abs().max() <= 1e-3 and (h1[:, -1] - dv[0, 0]).abs().max() <= 1e-3
    # the captured two-stream hipGraph replays the same bits
    gf = GraphedForward(model, imgs, proj, dv)
    rep = gf()
    torch.cuda.synchronize()
    for s in range(1, 5):
        for k in ("depth", "attn_weight", "photometric_confidence", "hypo_depth"):
            assert torch.equal(rep["stage%d" % s][k], out["stage%d" % s][k]), (s, k)
    del gf
    torch.cuda.empty_cache()


def _windows(h, w, wh=192, ww=224):
    """(y0, x0, y1, x1) windows of a stage map: far corner (largest offsets), near corner, an interior one; origins are
    multiples of 8 so that the U-Net's three stride-2 levels see the same sampling lattice as in the full map."""
    wh, ww = min(wh, h), min(ww, w)
    cands = {(h - wh, w - ww), (0, 0), (((h - wh) // 2) // 8 * 8, ((w - ww) // 3) // 8 * 8)}
    return [(y0, x0, y0 + wh, x0 + ww) for (y0, x0) in sorted(cands)]


def _interior(y0, x0, y1, x1, h, w, margin=48):
    """The part of a window whose U-Net receptive field (< 48 pixels) lies inside the window or beyond a true map border
    (where the window's zero padding is the real one)."""
    iy0 = 0 if y0 == 0 else margin
    ix0 = 0 if x0 == 0 else margin
    iy1 = (y1 - y0) if y1 == h else (y1 - y0) - margin
    ix1 = (x1 - x0) if x1 == w else (x1 - x0) - margin
    return slice(iy0, iy1), slice(ix0, ix1)


@pytest.mark.parametrize("H,W,N", FULL_SIZE)
def test_full_size_windows_vs_oracle(model, checkpoint, shipped_cfg, H, W, N):
    """Values at full size (large offsets, every tile edge class, 32-bit addressing) against the CPU oracle on windows
    of every stage: (a) the fused warp/correlation/aggregation kernel given the reference-style fp32 projection, (b)
    the regularisation U-Net given the kernel's own cost volume (window interior), (c) softmax / winner-take-all given
    the logits.  The oracle runs windows in seconds where the whole 1152x1600 map would take minutes."""
    from mvster_amd import ops
    from tests.test_gpu_kernels import _oracle_rt
    oracle = O.OracleMVS4net(**shipped_cfg)
    oracle.load_state_dict(checkpoint, strict=True)
    oracle.eval()
    imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=0)
    cap = {}
    out = model._forward_eval(*to_dev(imgs, proj, dv), capture=cap)
    worst = dict(cor_tight=0.0, cor_own=0.0, logits=0.0, attn=0.0, flips_clear=0.0)
    for s in range(4):
        name = "stage%d" % (s + 1)
        feats_cl = cap[name]["feats_cl"]                               # [N,1,h,w,C] on the GPU
        h, w, C = feats_cl.shape[2:]
        G, D = shipped_cfg["group_cor_dim"][s], shipped_cfg["stage_splits"][s]
        hypo = out[name]["hypo_depth"]
        pm = proj[name]
        # (a) same kernel, reference-style projection (fp32 torch.inverse on the CPU): pure per-pixel arithmetic
        rt_ref = _oracle_rt(pm).to(DEV)
        cor_tight = ops.warp_agg_fwd_cl(feats_cl[0].contiguous(), feats_cl[1:].contiguous(), rt_ref, hypo, G, True, True,
                                        2.0).permute(0, 4, 1, 2, 3)    # [1,G,D,h,w]
        cor_own = cap[name]["cor_feats"]
        feats = [feats_cl[v].permute(0, 3, 1, 2).cpu() for v in range(N)]    # NCHW on the CPU
        hypo_c = hypo.cpu()
        logits_gpu, attn_gpu, depth_gpu = cap[name]["logits"].cpu(), out[name]["attn_weight"].cpu(), out[name]["depth"].cpu()
        for (y0, x0, y1, x1) in _windows(h, w):
            with torch.no_grad():
                want = O.aggregate_views([feats[0][:, :, y0:y1, x0:x1]] + feats[1:], pm, hypo_c[:, :, y0:y1, x0:x1].contiguous(),
                                         True, G, attn_temp=2.0, attn_fuse_d=True, origin=(y0, x0))
                scale = max(want.abs().max().item(), 1.0)
                e_t = (cor_tight[:, :, :, y0:y1, x0:x1].cpu() - want).abs().max().item() / scale
                own_win = cor_own[:, :, :, y0:y1, x0:x1].cpu().contiguous()
                e_o = (own_win - want).abs().max().item() / scale
                worst["cor_tight"], worst["cor_own"] = max(worst["cor_tight"], e_t), max(worst["cor_own"], e_o)
                # one fp32 ulp of a sampling position is 1.2e-4 px at x ~ 1900 (the sgemm of rot @ pix may associate its
                # three terms either way): measured 4e-6 at 640 columns, 6.4e-5 in the far corner of 1920 columns;
                # an addressing or tile-edge bug shows up as O(1)
                assert e_t <= 2e-4, (name, (y0, x0), e_t)
                # (b) the U-Net on the kernel's own cost volume window
                lo = oracle.reg[s](own_win)                                # [1,D,wh,ww]
                iy, ix = _interior(y0, x0, y1, x1, h, w)
                lg = logits_gpu[:, :, y0:y1, x0:x1][:, :, iy, ix]
                lscale = max(lo.abs().max().item(), 1.0)
                e_l = (lg - lo[:, :, iy, ix]).abs().max().item() / lscale
                worst["logits"] = max(worst["logits"], e_l)
                assert e_l <= 5e-5, (name, (y0, x0), e_l)
                # (c) selection given the GPU's logits
                sel = O.select_depth(logits_gpu[:, :, y0:y1, x0:x1], hypo_c[:, :, y0:y1, x0:x1], s, True,
                                     shipped_cfg["depth_interals_ratio"][s])
                e_a = (attn_gpu[:, :, y0:y1, x0:x1] - sel["attn_weight"]).abs().max().item()
                worst["attn"] = max(worst["attn"], e_a)
                assert e_a <= 3e-7, (name, (y0, x0), e_a)
                top2 = sel["attn_weight"].topk(2, dim=1)[0]
                clear = (top2[:, 0] - top2[:, 1]) > 1e-6
                flips = (depth_gpu[:, y0:y1, x0:x1] != sel["depth"])[clear].float().mean().item()
                worst["flips_clear"] = max(worst["flips_clear"], flips)
                assert flips == 0.0, (name, (y0, x0), flips)
    note("full_size_windows_%dx%dx%d" % (H, W, N), **worst)
    # the kernel's own (fp64-inverse) projection differs from the reference's fp32 LAPACK inverse by that inverse's noise
    assert worst["cor_own"] <= 1e-3


@pytest.mark.parametrize("H,W,N", FULL_SIZE)
def test_full_size_fpn_windows_vs_oracle(model, checkpoint, shipped_cfg, H, W, N):
    """FPN4 at BASELINE sizes against the CPU oracle (models/mvs4net_utils.py:472-502): all four pyramid levels of the
    reference view and of the last source view on 384x448 image windows (far corner, origin, interior; origins multiples of
    8; a 96-pixel margin towards window edges that are not image edges), the oracle evaluating the align_corners
    up-sampling in whole-map coordinates (``FPN4.forward_window``, checked against the whole-map oracle on the CPU).
    Covers the fused fine-level kernels (fpn_tail_fused / fpn_tail_gather_lds / composed 3x3 layers) at the sizes the
    benchmark runs them at -- the small-size FPN tests stop at 128x192."""
    oracle = O.OracleMVS4net(**shipped_cfg)
    oracle.load_state_dict(checkpoint, strict=True)
    oracle.eval()
    imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=0)
    cap = {}
    model._forward_eval(*to_dev(imgs, proj, dv), capture=cap)
    wh, ww, margin = min(384, H), min(448, W), 96
    origins = sorted({(H - wh, W - ww), (0, 0), (((H - wh) // 2) // 8 * 8, ((W - ww) // 3) // 8 * 8)})
    worst = {"stage%d" % s: 0.0 for s in range(1, 5)}
    for v in (0, N - 1):
        for (y0, x0) in origins:
            with torch.no_grad():
                want = oracle.feature.forward_window(imgs[v][:, :, y0:y0 + wh, x0:x0 + ww].contiguous(), (y0, x0), (H, W))
            for s in range(1, 5):
                name, sc = "stage%d" % s, 2 ** (4 - s)
                got = cap[name]["feats_cl"][v, :, y0 // sc:(y0 + wh) // sc, x0 // sc:(x0 + ww) // sc].permute(0, 3, 1, 2).cpu()
                w_ = want[name]
                m = margin // sc
                iy = slice(0 if y0 == 0 else m, w_.shape[2] if y0 + wh == H else w_.shape[2] - m)
                ix = slice(0 if x0 == 0 else m, w_.shape[3] if x0 + ww == W else w_.shape[3] - m)
                scale = max(w_.abs().max().item(), 1.0)
                e = (got[:, :, iy, ix] - w_[:, :, iy, ix]).abs().max().item() / scale
                worst[name] = max(worst[name], e)
                assert e <= 5e-5, (name, v, (y0, x0), e)
    note("full_size_fpn_windows_%dx%dx%d" % (H, W, N), **worst)


def test_identical_views_have_uniform_attention_over_views(model):
    """If every source view equals the reference view (same image, same camera) the warp is the
    identity at every depth, so cor_feats = mean_c(f*f) for every depth and view."""
    imgs, proj, dv = make_inputs(nviews=3, H=64, W=128, seed=2)
    imgs = [imgs[0].clone() for _ in imgs]
    for k in proj:
        proj[k] = proj[k][:, :1].repeat(1, 3, 1, 1, 1).contiguous()
    imgs, proj, dv = to_dev(imgs, proj, dv)
    cap = {}
    out = model._forward_eval(imgs, proj, dv, capture=cap)
    f = out["stage4"]["mono_feat"]                                        # [1,8,H,W]
    want = (f * f).reshape(1, 4, 2, 64, 128).mean(2)                      # G=4
    got = cap["stage4"]["cor_feats"]                                      # [1,4,D,H,W]
    inner = (slice(None), slice(None), slice(2, -2), slice(2, -2))
    for d in range(got.shape[2]):
        assert (got[:, :, d][inner] - want[inner]).abs().max() <= 2e-4 * want.abs().max()


def test_train_step_vs_golden(shipped_cfg, checkpoint, golden):
    """Train-mode forward + OT loss + backward (all HIP: convolutions, BatchNorm, warp fwd/bwd, Sinkhorn) vs the reference."""
    g = golden("g6_train")
    H, W, N = int(g.np("H")), int(g.np("W")), int(g.np("N"))
    imgs, proj, dv = to_dev(*make_inputs(nviews=N, H=H, W=W, seed=2, batch=2))
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).train()
    out = m(imgs, proj, dv)
    gt = {"stage%d" % s: g.t("depth_gt_stage%d" % s, DEV) for s in range(1, 5)}
    mask = {"stage%d" % s: g.t("mask_stage%d" % s, DEV) for s in range(1, 5)}
    loss, l1s, ots, _ = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True,
                                     ot_iter=10, ot_eps=1, ot_continous=False, mono=True)
    loss.backward()
    torch.cuda.synchronize()
    want_loss = float(g.np("loss"))
    a1 = (out["stage1"]["attn_weight"].detach().cpu() - g.t("stage1_attn_weight")).abs().max().item()
    named = dict(m.named_parameters())
    worst = 0.0
    for k in [k for k in g.keys() if k.startswith("grad/")]:
        ref = g.t(k)
        rel = ((named[k[5:]].grad.cpu() - ref).abs().max() / (ref.abs().max() + 1e-12)).item()
        worst = max(worst, rel)
    note("train_step", loss=loss.item(), want_loss=want_loss, stage1_attn_max=a1, worst_grad_rel=worst)
    assert out["photometric_confidence"].dim() == 0
    assert a1 <= 1e-3
    # the reference's own parameter gradients (7 tensors across the FPN and the U-Nets); measured 0.86 % (near-tie argmax
    # flips against the reference move individual hypotheses; the per-module gradient tests in test_gpu_train.py, <= 4e-6
    # relative, are what pins the kernels)
    assert worst <= 1.5e-2
    # later stages can pick other hypotheses on near-ties, so the total loss is compared loosely
    assert abs(ots[0].item() - float(g.np("ot")[0])) <= 2e-3 * abs(float(g.np("ot")[0]))
    assert abs(loss.item() - want_loss) <= 5e-2 * abs(want_loss)
    for k in named:
        if named[k].grad is not None:
            assert torch.isfinite(named[k].grad).all(), k


@pytest.mark.parametrize("name,kw", [
    ("reg3d_inverse_group", dict(reg_net="reg3d", group_cor=True, group_cor_dim=[8, 8, 4, 4], inverse_depth=True, mono=False)),
    ("reg2d_linear_depth", dict(reg_net="reg2d", group_cor=True, group_cor_dim=[8, 8, 4, 4], inverse_depth=False, mono=True)),
    ("reg2d_sqdiff_nofuse", dict(reg_net="reg2d", group_cor=False, inverse_depth=True, mono=False, attn_fuse_d=False,
                                 attn_temp=1)),
    # a free --ndepths (the reference's schedulers and stagenet take any count, mvs4net_utils.py:61-99,1012-1094): more
    # hypotheses than the fused forward kernels hold in registers (general warp kernel with 32 / 16 pixels per workgroup,
    # memory-walking selection), odd counts, and the two-hypothesis minimum of the linear-depth cascade
    ("reg2d_inverse_48_32_8_4", dict(reg_net="reg2d", group_cor=True, group_cor_dim=[8, 8, 4, 4], inverse_depth=True, mono=True,
                                     stage_splits=[48, 32, 8, 4])),
    ("reg2d_linear_24_17_2_3", dict(reg_net="reg2d", group_cor=True, group_cor_dim=[8, 8, 4, 4], inverse_depth=False, mono=False,
                                    stage_splits=[24, 17, 2, 3])),
    ("reg2d_sqdiff_8_8_64_4", dict(reg_net="reg2d", group_cor=False, inverse_depth=True, mono=False, attn_fuse_d=False,
                                   attn_temp=1, stage_splits=[8, 8, 64, 4])),
])
def test_other_configurations_vs_oracle(name, kw):
    """Options outside the shipped script (reg3d, linear-depth schedulers, squared-difference volume,
    attn_fuse_d=False, other hypothesis counts): the HIP eval path against the CPU oracle with the same weights,
    teacher-forced."""
    from mvster_amd.synthetic import randomize_state
    cfg = dict(arch_mode="fpn", num_stage=4, fpn_base_channel=8, reg_channel=8, stage_splits=[8, 8, 4, 4],
               depth_interals_ratio=[0.5, 0.5, 0.5, 1], attn_temp=2, attn_fuse_d=True)
    cfg.update(kw)
    torch.manual_seed(3)
    oracle = O.OracleMVS4net(**cfg)
    sd = randomize_state(oracle.state_dict(), seed=11, prob_gain=20.0)
    oracle.load_state_dict(sd, strict=True)
    oracle.eval()
    m = MVS4net(**cfg)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).eval()
    H, W, N = 128, 192, 3
    imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=6)
    cap_o = {}
    with torch.no_grad():
        want = oracle(imgs, proj, dv, capture=cap_o)
    teacher = {"stage%d" % s: want["stage%d" % s]["hypo_depth"].to(DEV) for s in range(1, 5)}
    cap = {}
    got = m._forward_eval(*to_dev(imgs, proj, dv), teacher=teacher, capture=cap)
    worst_cor = worst_attn = 0.0
    for s in range(1, 5):
        st, wt = got["stage%d" % s], want["stage%d" % s]
        wc = cap_o["stage%d" % s]["cor_feats"]
        worst_cor = max(worst_cor, ((cap["stage%d" % s]["cor_feats"].cpu() - wc).abs().max() / wc.abs().max()).item())
        worst_attn = max(worst_attn, (st["attn_weight"].cpu() - wt["attn_weight"]).abs().max().item())
        top2 = wt["attn_weight"].topk(2, dim=1)[0]
        clear = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert (st["depth"].cpu() - wt["depth"])[clear].abs().mean() < 1e-4, (name, s)
        assert set(st.keys()) == set(wt.keys()), (name, s)
    note("cfg_" + name, cor_rel_max=worst_cor, attn_max=worst_attn)
    assert worst_cor <= 2e-4 and worst_attn <= 1e-3
    # the free-running cascade (own schedulers, incl. the linear-depth ones) stays finite and consistent
    out = m(*to_dev(imgs, proj, dv))
    for s in range(1, 5):
        assert torch.isfinite(out["stage%d" % s]["depth"]).all()
    if not cfg["inverse_depth"]:
        s2 = out["stage2"]["hypo_depth"].cpu()
        with torch.no_grad():
            ref_h = O.schedule_range(out["stage1"]["depth"].cpu(), cfg["stage_splits"][1], 0.5 * (dv[:, -1] - dv[:, 0]) / dv.size(1), 32, 48)
        assert (s2 - ref_h).abs().max() <= 5e-7 * ref_h.abs().max()


def _all_leaves(out):
    for k, v in out.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                yield (k, kk), vv
        else:
            yield (k,), v


def _clone_out(out):
    return {k: ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v.clone()) for k, v in out.items()}


def test_forward_graph_cache_is_transparent(shipped_cfg, checkpoint):
    """``MVS4net.forward`` itself replays a captured hipGraph from the second call of a shape on (graph.ForwardCache): the
    unchanged ``model(imgs, proj, depth_values)`` loop of the reference's drivers (test_mvs4.py:202-207).  Bit-identical to
    the eager forward on every key of every stage, outputs freshly allocated per call (the reference's contract), the
    flattened last-stage entries the same tensors as ``stage4``'s, same key order."""
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    a = to_dev(*make_inputs(nviews=5, H=128, W=192, seed=21))
    b = to_dev(*make_inputs(nviews=5, H=128, W=192, seed=22))
    want_a, want_b = _clone_out(m.forward_eager(*a)), _clone_out(m.forward_eager(*b))
    o1 = m(*a)                                   # first sight: eager
    assert m._fwd_cache.stats["eager"] == 1 and m._fwd_cache.stats["captured"] == 0
    o2 = m(*b)                                   # second call of the shape: capture, replay on b
    o3 = m(*a)                                   # replay on a
    torch.cuda.synchronize()
    assert m._fwd_cache.stats == {"eager": 1, "captured": 1, "replayed": 2, "capture_failed": 0}
    assert list(o3.keys()) == list(want_a.keys()) and list(o3["stage2"].keys()) == list(want_a["stage2"].keys())
    for got, want in ((o1, want_a), (o2, want_b), (o3, want_a)):      # (o2 is checked AFTER o3 was produced: not overwritten)
        for path, t in _all_leaves(got):
            w_ = want[path[0]] if len(path) == 1 else want[path[0]][path[1]]
            assert t.shape == w_.shape and torch.equal(t, w_), path
    ptr2 = {t.data_ptr() for _, t in _all_leaves(o2)}
    ptr3 = {t.data_ptr() for _, t in _all_leaves(o3)}
    assert not (ptr2 & ptr3)                      # fresh storage per call
    for k in ("depth", "photometric_confidence", "attn_weight", "hypo_depth", "inverse_min_depth", "mono_feat"):
        assert o3[k] is o3["stage4"][k], k        # the flattening of MVS4Net.py:104-105 keeps identity
    assert tuple(o3["mono_feat"].shape) == (1, 8, 128, 192)
    # host-side projection matrices in float64 (a caller that skips tocuda for the small tensors): converted like the eager path
    proj64 = {k: v.double().cpu() for k, v in a[1].items()}
    o4 = m(a[0], proj64, a[2])
    assert torch.equal(o4["depth"], want_a["depth"]) and torch.equal(o4["stage1"]["attn_weight"], want_a["stage1"]["attn_weight"])
    # switched off: every call eager again
    m.graph_cache = False
    before = dict(m._fwd_cache.stats)
    assert torch.equal(m(*b)["depth"], want_b["depth"]) and m._fwd_cache.stats == before


def test_forward_graph_cache_follows_weight_updates_shapes_and_mode_changes(shipped_cfg, checkpoint):
    """A cached graph is replayed only while the parameters / buffers it was folded from are unchanged: an in-place update
    sends the next call back to the eager path (fresh plans), the call after that re-captures.  A new shape gets its own
    entry and the old one stays valid.  A train()/eval() round trip without a parameter change rebuilds the model's plans
    while the captured entry keeps (and owns) the ones it recorded."""
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    a = to_dev(*make_inputs(nviews=3, H=64, W=128, seed=5))
    c = to_dev(*make_inputs(nviews=3, H=128, W=128, seed=6))
    m(*a)
    before = m(*a)["stage4"]["attn_weight"]
    assert m._fwd_cache.stats["captured"] == 1
    with torch.no_grad():
        m.reg[3].conv0.conv.weight.mul_(1.5)
        m.feature.conv0[0].bn.running_mean.add_(0.05)
    fresh = MVS4net(**shipped_cfg)
    fresh.load_state_dict(m.state_dict(), strict=True)
    fresh.to(DEV).eval()
    want = fresh.forward_eager(*a)["stage4"]["attn_weight"]
    after1 = m(*a)["stage4"]["attn_weight"]                        # eager on the new weights
    after2 = m(*a)["stage4"]["attn_weight"]                        # re-captured
    after3 = m(*a)["stage4"]["attn_weight"]
    st = m._fwd_cache.stats
    assert (st["eager"], st["captured"]) == (2, 2) and st["replayed"] == 3
    assert torch.equal(after1, want) and torch.equal(after2, want) and torch.equal(after3, want) and not torch.equal(before, want)
    # another shape: its own entry; the first shape still replays
    want_c = fresh.forward_eager(*c)["depth"]
    assert torch.equal(m(*c)["depth"], want_c) and torch.equal(m(*c)["depth"], want_c)
    assert torch.equal(m(*a)["stage4"]["attn_weight"], want) and m._fwd_cache.stats["captured"] == 3
    assert len(m._fwd_cache.entries) == 2
    # train() / eval() round trip, parameters untouched: plans rebuilt, the cached graph keeps its own and stays valid
    m.train()
    m.eval()
    assert torch.equal(m(*a)["stage4"]["attn_weight"], want) and m._fwd_cache.stats["captured"] == 3
    # load_state_dict drops the cache (the plans it recorded are folded from other weights)
    m.load_state_dict(checkpoint, strict=True)
    assert len(m._fwd_cache.entries) == 0
    assert torch.equal(m(*a)["stage4"]["attn_weight"], before)


def test_forward_graph_cache_sees_batchnorm_statistics_moved_by_a_training_forward(shipped_cfg, checkpoint):
    """The native BatchNorm writes its running statistics through raw pointers (no ``_version`` bump): a train-mode forward
    WITHOUT an optimizer step (BatchNorm re-calibration, a frozen-backbone fine-tune, a no_grad forward in training mode)
    must still send the next eval call back to fresh plans instead of replaying the graph folded from the old statistics."""
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    a = to_dev(*make_inputs(nviews=3, H=64, W=128, seed=7, batch=2))
    m(*a)
    before = m(*a)["stage4"]["attn_weight"].clone()
    assert m._fwd_cache.stats["captured"] == 1
    rm0 = m.feature.conv0[0].bn.running_mean.clone()
    m.train()
    with torch.no_grad():
        m(*a)                                                        # moves the running statistics, nothing else
    m.eval()
    assert not torch.equal(rm0, m.feature.conv0[0].bn.running_mean)
    got = m(*a)["stage4"]["attn_weight"]
    want = m.forward_eager(*a)["stage4"]["attn_weight"]
    assert torch.equal(got, want) and not torch.equal(got, before)
    assert torch.equal(m(*a)["stage4"]["attn_weight"], want) and m._fwd_cache.stats["captured"] == 2


def test_outputs_to_numpy_equals_tensor2numpy(shipped_cfg, checkpoint):
    """graph.outputs_to_numpy -- the reference's tensor2numpy(outputs) (utils.py:50-57) with one device -> host copy for a
    graph-cache result -- returns the same arrays, in the reference's structure, freshly allocated; ``keys`` keeps a subset."""
    from mvster_amd.graph import outputs_to_numpy
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    a = to_dev(*make_inputs(nviews=3, H=64, W=128, seed=9))
    for call in range(3):                                            # eager result, captured, replayed
        o = m(*a)
        want = {k: ({k2: v2.detach().cpu().numpy().copy() for k2, v2 in v.items()} if isinstance(v, dict) else v.detach().cpu().numpy().copy())
                for k, v in o.items()}
        got = outputs_to_numpy(o)
        assert list(got.keys()) == list(want.keys())
        for k, v in want.items():
            if isinstance(v, dict):
                assert list(got[k].keys()) == list(v.keys())
                for k2, w2 in v.items():
                    assert got[k][k2].shape == w2.shape and np.array_equal(got[k][k2], w2), (call, k, k2)
            else:
                assert got[k].shape == v.shape and np.array_equal(got[k], v), (call, k)
        sub = outputs_to_numpy(o, keys=("depth", "photometric_confidence"))
        assert sorted(sub["stage2"].keys()) == ["depth", "photometric_confidence"] and np.array_equal(sub["depth"], want["depth"])
    second = outputs_to_numpy(m(*a))
    assert second["depth"] is not got["depth"] and np.array_equal(second["depth"], got["depth"])     # not the staging buffer


def test_fused_hypothesis_scheduling_gives_the_same_forward(shipped_cfg, checkpoint):
    """``MVS4net.fuse_hypotheses`` (off by default: measured slower): every stage's hypotheses computed inside the warp launch
    (mvster_warp_agg_fwd_sched) -- the same forward, bit for bit, eager and through the graph cache (the switch is part of
    the cache key)."""
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    a = to_dev(*make_inputs(nviews=4, H=128, W=192, seed=31, batch=2))
    want = _clone_out(m.forward_eager(*a))
    m.fuse_hypotheses = True
    for got in (m(*a), m(*a), m(*a)):                     # eager, captured, replayed
        for path, t in _all_leaves(got):
            w_ = want[path[0]] if len(path) == 1 else want[path[0]][path[1]]
            assert torch.equal(t, w_), path
    assert m._fwd_cache.stats["captured"] == 1
    m.fuse_hypotheses = False
    assert torch.equal(m(*a)["depth"], want["depth"]) and m._fwd_cache.stats["eager"] == 2      # another key: eager first


@pytest.mark.parametrize("num_stage", [4, 2])
def test_merged_launches_give_the_same_forward(shipped_cfg, checkpoint, num_stage):
    """``MVS4net.merge_launches`` (default): pack + projections + first hypotheses in one launch, the coarse stages' confidence
    up-samplings in one -- the forward of the separate launches, bit for bit, for every output of every stage."""
    cfg = dict(shipped_cfg)
    if num_stage != 4:
        cfg.update(num_stage=num_stage, stage_splits=shipped_cfg["stage_splits"][:num_stage],
                   group_cor_dim=shipped_cfg["group_cor_dim"][:num_stage],
                   depth_interals_ratio=shipped_cfg["depth_interals_ratio"][:num_stage])
    m = MVS4net(**cfg)
    if num_stage == 4:
        m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    a = to_dev(*make_inputs(nviews=4, H=128, W=192, seed=37, batch=2))
    assert m.merge_launches
    m.merge_launches = False
    want = _clone_out(m.forward_eager(*a))
    m.merge_launches = True
    for got in (m(*a), m(*a), m(*a)):                     # eager, captured, replayed
        n = 0
        for path, t in _all_leaves(got):
            w_ = want[path[0]] if len(path) == 1 else want[path[0]][path[1]]
            assert t.shape == w_.shape and torch.equal(t, w_), path
            n += 1
        assert n == sum(1 for _ in _all_leaves(want))
    for k in range(num_stage):
        assert got["stage%d" % (k + 1)]["photometric_confidence"].shape == (2, 128, 192)


def test_graphed_forward_refuses_stale_weights(shipped_cfg, checkpoint):
    """``GraphedForward`` records the model's state stamp at capture: after an in-place parameter update its ``__call__``
    raises instead of replaying the weights folded at capture time."""
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    a = to_dev(*make_inputs(nviews=3, H=64, W=128, seed=5))
    gf = GraphedForward(m, *a)
    gf()
    with torch.no_grad():
        m.reg[0].prob.bias.add_(0.5)
    with pytest.raises(RuntimeError, match="changed after the capture"):
        gf()
    gf(check_state=False)                                             # (the raw replay stays available)
    torch.cuda.synchronize()


def test_under_data_parallel(shipped_cfg, checkpoint):
    """The reference's test driver wraps the model in nn.DataParallel (test_mvs4.py:196) and indexes the outputs
    with tensor2numpy: same results through the wrapper, every leaf a tensor."""
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    imgs, proj, dv = to_dev(*make_inputs(nviews=3, H=128, W=192, seed=9))
    want = m(imgs, proj, dv)
    dp = torch.nn.DataParallel(m)
    dp.eval()
    got = dp(imgs, proj, dv)
    got = dp(imgs, proj, dv)               # (second call of the shape through the wrapper: MVS4net.forward's graph cache replays)
    assert m._fwd_cache.stats["replayed"] >= 1
    assert set(got.keys()) == set(want.keys())
    for k in ("depth", "photometric_confidence", "attn_weight", "hypo_depth"):
        assert torch.equal(got[k], want[k]), k
    for s in range(1, 5):
        for k, v in got["stage%d" % s].items():
            assert isinstance(v, torch.Tensor), (s, k)
    state = dp.state_dict()
    assert all(k.startswith("module.") for k in state) and len(state) == len(m.state_dict())


@pytest.mark.parametrize("B,N", [(2, 2), (1, 7), (1, 9), (3, 4)])
def test_batch_and_view_counts_vs_oracle(shipped_cfg, checkpoint, B, N):
    """Batch sizes and view counts other than the benchmark's (one source view; nine views; batch 3): shipped
    configuration, teacher-forced against the CPU oracle."""
    oracle = O.OracleMVS4net(**shipped_cfg)
    oracle.load_state_dict(checkpoint, strict=True)
    oracle.eval()
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    imgs, proj, dv = make_inputs(nviews=N, H=64, W=128, seed=B * 10 + N, batch=B)
    with torch.no_grad():
        want = oracle(imgs, proj, dv)
    teacher = {"stage%d" % s: want["stage%d" % s]["hypo_depth"].to(DEV) for s in range(1, 5)}
    got = m._forward_eval(*to_dev(imgs, proj, dv), teacher=teacher)
    for s in range(1, 5):
        st, wt = got["stage%d" % s], want["stage%d" % s]
        assert tuple(st["depth"].shape) == tuple(wt["depth"].shape)
        assert (st["attn_weight"].cpu() - wt["attn_weight"]).abs().max() <= 1e-3, (B, N, s)
        top2 = wt["attn_weight"].topk(2, dim=1)[0]
        clear = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert (st["depth"].cpu() - wt["depth"])[clear].abs().mean() < 1e-4, (B, N, s)
    assert tuple(got["photometric_confidence"].shape) == (B, 64, 128)


def test_batched_forward_equals_one_map_per_call_full_size(shipped_cfg, checkpoint):
    """Four depth maps per forward call at the benchmark's size (bench.py `value_batched`: the batch shares the coarse stages'
    short launches) against the same four maps one per call: each stage teacher-forced with the hypotheses of the single
    calls.  The batched shapes pick other kernels here and there (family fallback of the table: Winograd or direct forms sum
    in different orders), so: attention within 2e-4, the selected depth identical wherever the two leading probabilities are
    1e-3 apart, and the monocular features of the reference views within 1e-5 of their scale."""
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    B, N, H, W = 4, 5, 512, 640
    imgs, proj, dv = to_dev(*make_inputs(nviews=N, H=H, W=W, seed=31, batch=B))
    singles = []
    for b in range(B):
        singles.append(m([i[b:b + 1] for i in imgs], {k: v[b:b + 1] for k, v in proj.items()}, dv[b:b + 1]))
    teacher = {"stage%d" % s: torch.cat([o["stage%d" % s]["hypo_depth"] for o in singles]) for s in range(1, 5)}
    got = m._forward_eval(imgs, proj, dv, teacher=teacher)
    worst = 0.0
    for s in range(1, 5):
        st = got["stage%d" % s]
        want_attn = torch.cat([o["stage%d" % s]["attn_weight"] for o in singles])
        want_depth = torch.cat([o["stage%d" % s]["depth"] for o in singles])
        worst = max(worst, (st["attn_weight"] - want_attn).abs().max().item())
        top2 = want_attn.topk(2, dim=1)[0]
        clear = (top2[:, 0] - top2[:, 1]) > 1e-3
        assert torch.equal(st["depth"][clear], want_depth[clear]), s
        assert clear.float().mean() > 0.3           # (random weights: flat distributions; the check must not be vacuous)
        if "mono_feat" in st:
            wm = torch.cat([o["stage%d" % s]["mono_feat"] for o in singles])
            assert (st["mono_feat"] - wm).abs().max() <= 1e-5 * wm.abs().max()
    note("batched_forward_B4", attn_max=worst)
    assert worst <= 2e-4
    assert tuple(got["depth"].shape) == (B, H, W)


def test_stage1_only_configuration_vs_oracle():
    """BASELINE config 1: DTU mid 512x640, 5 views, stage-1 only (8 hypotheses), B=1 -- the whole forward against the
    CPU oracle at full size (the FPN of five views plus one 64x80 stage is cheap enough for the CPU)."""
    from mvster_amd.synthetic import randomize_state
    cfg = dict(arch_mode="fpn", reg_net="reg2d", num_stage=1, fpn_base_channel=8, reg_channel=8, stage_splits=[8],
               depth_interals_ratio=[0.5], group_cor=True, group_cor_dim=[8], inverse_depth=True, mono=False, attn_temp=2,
               attn_fuse_d=True)
    torch.manual_seed(8)
    oracle = O.OracleMVS4net(**cfg)
    sd = randomize_state(oracle.state_dict(), seed=13, prob_gain=20.0)
    oracle.load_state_dict(sd, strict=True)
    oracle.eval()
    m = MVS4net(**cfg)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).eval()
    imgs, proj, dv = make_inputs(nviews=5, H=512, W=640, seed=12)
    cap_o, cap = {}, {}
    with torch.no_grad():
        want = oracle(imgs, proj, dv, capture=cap_o)
    got = m._forward_eval(*to_dev(imgs, proj, dv), capture=cap)
    assert set(got.keys()) == set(want.keys()) == {"stage1", "depth", "photometric_confidence", "hypo_depth", "attn_weight",
                                                   "inverse_min_depth", "inverse_max_depth"}
    assert torch.equal(got["hypo_depth"].cpu(), want["hypo_depth"])
    wc = cap_o["stage1"]["cor_feats"]
    e_cor = ((cap["stage1"]["cor_feats"].cpu() - wc).abs().max() / wc.abs().max()).item()
    e_attn = (got["attn_weight"].cpu() - want["attn_weight"]).abs().max().item()
    top2 = want["attn_weight"].topk(2, dim=1)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3
    l1 = (got["depth"].cpu() - want["depth"])[clear].abs().mean().item()
    conf = (got["photometric_confidence"].cpu() - want["photometric_confidence"]).abs().max().item()
    note("stage1_only_512x640x5", cor_rel_max=e_cor, attn_max=e_attn, depth_l1_clear=l1, conf_max=conf,
         clear_frac=clear.float().mean().item())
    assert tuple(got["photometric_confidence"].shape) == (1, 512, 640)
    assert e_cor <= 2e-4 and e_attn <= 1e-3 and l1 < 1e-4 and conf <= 1e-3
    # the public call (own hypotheses, graph capture) gives the same result
    out = m(*to_dev(imgs, proj, dv))
    assert torch.equal(out["depth"], got["depth"])


def test_train_step_full_size_vs_pytorch_rocm(shipped_cfg, checkpoint):
    """BASELINE config 4 on one rank at full size: 512x640, 5 views, B=2, one training step (forward, OT loss, backward)
    of the native path against the same step of the oracle module tree on the GPU (plain PyTorch-ROCm)."""
    import time
    from mvster_amd import MVS4net_loss
    H, W, N, B = 512, 640, 5, 2
    ref = O.OracleMVS4net(**shipped_cfg)
    ref.load_state_dict(checkpoint, strict=True)
    nat = MVS4net(**shipped_cfg)
    nat.load_state_dict(checkpoint, strict=True)
    ref.to(DEV).train()
    nat.to(DEV).train()
    imgs, proj, dv = to_dev(*make_inputs(nviews=N, H=H, W=W, seed=21, batch=B))
    g = torch.Generator().manual_seed(0)
    gt, mask = {}, {}
    for s in range(1, 5):
        hs, ws = H // 2 ** (4 - s), W // 2 ** (4 - s)
        gt["stage%d" % s] = (500 + 300 * torch.rand(B, hs, ws, generator=g)).to(DEV)
        mask["stage%d" % s] = (torch.rand(B, hs, ws, generator=g) > 0.2).float().to(DEV)
    kw = dict(stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)

    def step(m, loss_fn, images):
        m.zero_grad(set_to_none=True)
        out = m(images, proj, dv)
        res = loss_fn(out, gt, mask, **kw)
        res[0].backward()
        torch.cuda.synchronize()
        return out, res, {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    t0 = time.perf_counter()
    o_ref, r_ref, g_ref = step(ref, O.mvs4net_loss, imgs)
    t_ref = time.perf_counter() - t0
    # yardstick: the PyTorch-ROCm step with the images perturbed by 1e-6 relative (the step is ill-conditioned: argmax
    # depths feed the next stage's hypotheses, the OT loss takes logs of small probabilities)
    gp = torch.Generator().manual_seed(11)
    pert = [i * (1 + 1e-6 * torch.randn(i.shape, generator=gp).to(DEV)) for i in imgs]
    bufs = {k: v.detach().clone() for k, v in ref.named_buffers()}
    _, _, g_ref2 = step(ref, O.mvs4net_loss, pert)
    t0 = time.perf_counter()
    o_nat, r_nat, g_nat = step(nat, MVS4net_loss, imgs)
    t_nat = time.perf_counter() - t0
    a1 = (o_ref["stage1"]["attn_weight"] - o_nat["stage1"]["attn_weight"]).abs().max().item()
    ot1 = abs(r_ref[2][0].item() - r_nat[2][0].item()) / abs(r_ref[2][0].item())
    l_ref, l_nat = r_ref[0].item(), r_nat[0].item()
    gmax = max(v.norm().item() for v in g_ref.values())
    worst, worst_name, noise = 0.0, "", 0.0
    for k, r in g_ref.items():
        if r.norm().item() < 1e-4 * gmax:
            continue
        e = ((g_nat[k] - r).norm() / r.norm()).item()
        noise = max(noise, ((g_ref2[k] - r).norm() / r.norm()).item())
        if e > worst:
            worst, worst_name = e, k
    note("train_step_full_size_512x640x5_B2", loss_ref=l_ref, loss_native=l_nat, stage1_attn_max=a1, stage1_ot_rel=ot1,
         worst_grad_rel_l2=worst, pytorch_path_1e6_perturbation_rel_l2=noise, first_step_s_pytorch_rocm=t_ref,
         first_step_s_native=t_nat)
    assert set(g_ref) == set(g_nat)
    assert a1 <= 1e-3 and ot1 <= 2e-3          # measured 2.7e-4 / 1e-7 (the small-size step: 3e-5)
    assert abs(l_ref - l_nat) <= 2e-2 * abs(l_ref)
    assert worst <= max(2e-2, 1.25 * noise), (worst_name, worst, noise)     # measured 13 % at 12 % noise
    for k, v in g_nat.items():
        assert torch.isfinite(v).all(), k


def _train_step_full_size_teacher_forced(shipped_cfg, checkpoint, tag, well_conditioned):
    from mvster_amd import MVS4net_loss
    checkpoint = dict(checkpoint)
    if well_conditioned:
        gq = torch.Generator().manual_seed(77)
        for i in range(4):            # nn.Conv3d(8, 1, 1) default initialisation: U(-1/sqrt(8), 1/sqrt(8)) for weight and bias
            bound = 8 ** -0.5
            checkpoint["reg.%d.prob.weight" % i] = (torch.rand(1, 8, 1, 1, 1, generator=gq) * 2 - 1) * bound
            checkpoint["reg.%d.prob.bias" % i] = (torch.rand(1, generator=gq) * 2 - 1) * bound
    H, W, N, B = 512, 640, 5, 2
    ref = O.OracleMVS4net(**shipped_cfg)
    ref.load_state_dict(checkpoint, strict=True)
    nat = MVS4net(**shipped_cfg)
    nat.load_state_dict(checkpoint, strict=True)
    ref.to(DEV).train()
    nat.to(DEV).train()
    imgs, proj, dv = to_dev(*make_inputs(nviews=N, H=H, W=W, seed=21, batch=B))
    g = torch.Generator().manual_seed(0)
    gt, mask = {}, {}
    for s in range(1, 5):
        hs, ws = H // 2 ** (4 - s), W // 2 ** (4 - s)
        gt["stage%d" % s] = (500 + 300 * torch.rand(B, hs, ws, generator=g)).to(DEV)
        mask["stage%d" % s] = (torch.rand(B, hs, ws, generator=g) > 0.2).float().to(DEV)
    kw = dict(stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)
    with torch.no_grad():
        teacher = {k: v["hypo_depth"].clone() for k, v in ref(imgs, proj, dv).items() if isinstance(v, dict)}

    def ref_step(images):
        ref.load_state_dict(checkpoint, strict=True)       # (every forward moves BatchNorm's running statistics)
        ref.zero_grad(set_to_none=True)
        out = ref(images, proj, dv, teacher=teacher)
        res = O.mvs4net_loss(out, gt, mask, **kw)
        res[0].backward()
        torch.cuda.synchronize()
        return ({k: out[k]["attn_weight"].detach().clone() for k in teacher}, res[0].item(),
                {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None})

    a_ref, l_ref, g_ref = ref_step(imgs)
    a_again, _, g_again = ref_step(imgs)                   # run-to-run (atomics)
    gp = torch.Generator().manual_seed(11)
    a_pert, _, g_pert = ref_step([i * (1 + 1e-6 * torch.randn(i.shape, generator=gp).to(DEV)) for i in imgs])
    nat.zero_grad(set_to_none=True)
    nat._check_inputs(imgs, proj, dv)
    o_nat = nat._forward_train(imgs, proj, dv, teacher=teacher)
    r_nat = MVS4net_loss(o_nat, gt, mask, **kw)
    r_nat[0].backward()
    torch.cuda.synchronize()
    l_nat = r_nat[0].item()
    g_nat = {k: p.grad for k, p in nat.named_parameters() if p.grad is not None}
    assert set(g_ref) == set(g_nat)
    attn = {k: (a_ref[k] - o_nat[k]["attn_weight"]).abs().max().item() for k in teacher}
    attn_noise = {k: max((a_ref[k] - a_pert[k]).abs().max().item(), (a_ref[k] - a_again[k]).abs().max().item()) for k in teacher}
    gmax = max(v.norm().item() for v in g_ref.values())
    rows, num, den, ynum = [], 0.0, 0.0, 0.0
    for k, r in g_ref.items():
        rn = r.norm().item()
        e = (g_nat[k] - r).norm().item()
        num, den = num + e * e, den + rn * rn
        ynum += max((g_again[k] - r).norm().item(), (g_pert[k] - r).norm().item()) ** 2
        if rn < 1e-4 * gmax:
            assert e <= 1e-6 * gmax, (k, e, gmax)          # a near-zero gradient: measured against the largest one
            continue
        yard = max((g_again[k] - r).norm().item(), (g_pert[k] - r).norm().item()) / rn
        rows.append((e / rn, yard, k))
    overall, overall_yard = (num / den) ** 0.5, (ynum / den) ** 0.5
    tight = [r for r in rows if r[1] <= 1e-3]              # tensors PyTorch itself reproduces to 1e-3
    worst_tight = max(tight) if tight else (0.0, 0.0, "")
    worst_any = max(rows)
    excess = max(rows, key=lambda r: r[0] / max(8 * r[1], 2e-3))
    note("train_step_full_size_teacher_forced_512x640x5_B2" + tag, loss_ref=l_ref, loss_native=l_nat,
         attn_max=max(attn.values()), attn_max_pytorch_vs_itself=max(attn_noise.values()),
         all_parameters_grad_rel_l2=overall, all_parameters_pytorch_yardstick=overall_yard, tensors=len(rows), tensors_pytorch_reproduces_to_1e3=len(tight),
         worst_grad_rel_l2_among_those=worst_tight[0], worst_grad_rel_l2_any=worst_any[0], its_pytorch_yardstick=worst_any[1],
         median_grad_rel_l2=sorted(r[0] for r in rows)[len(rows) // 2],
         median_pytorch_yardstick=sorted(r[1] for r in rows)[len(rows) // 2])
    print("teacher-forced full-size step: loss %.6f / %.6f; attn %s (PyTorch vs itself %s); all parameters %.2e; %d of %d tensors "
          "reproducible to 1e-3 by PyTorch, worst of those %s %.2e; worst of all %s %.2e (yardstick %.2e)"
          % (l_ref, l_nat, {k: "%.1e" % v for k, v in attn.items()}, {k: "%.1e" % v for k, v in attn_noise.items()}, overall,
             len(tight), len(rows), worst_tight[2], worst_tight[0], worst_any[2], worst_any[0], worst_any[1]))
    assert abs(l_ref - l_nat) <= 1e-5 * abs(l_ref)
    if well_conditioned:
        # default-initialised prob heads: no saturated softmax in front of the OT loss's logarithms, PyTorch reproduces
        # itself -- and then every parameter gradient is asked to 1e-3 (where PyTorch's own yardstick is below 1e-4; a
        # tensor PyTorch itself moves more than that keeps the 8 x yardstick rule)
        #  Measured: the attention volumes now agree to 1e-5 .. 2.7e-4 (PyTorch against itself 1.5e-5 .. 3.7e-5), but the
        #  gradients stay ill-conditioned with ANY heads -- PyTorch reproduces 28 of the 167 tensors to 1e-3 (the random
        #  ground truth puts most pixels' mass far from the prediction: the OT loss works on log(1e-12 + p)) -- so the
        #  whole-vector bound is again relative to PyTorch's own yardstick.
        bad = [(e, y, k) for e, y, k in rows if e > (1e-3 if y <= 1e-4 else max(1e-3, 8 * y))]
        very_tight = [r for r in rows if r[1] <= 1e-4]
        note("train_step_full_size_teacher_forced_512x640x5_B2" + tag + "_rule", tensors_pytorch_reproduces_to_1e4=len(very_tight),
             worst_of_those=max(very_tight)[0] if very_tight else 0.0, tensors_over_their_bound=len(bad))
        assert not bad, sorted(bad)[-5:]
        assert overall <= max(1e-3, 1.5 * overall_yard), (overall, overall_yard)
        return
    for k in teacher:
        # (measured 2.5e-4 .. 4.6e-3 from stage 1 to 4 against 8e-4 .. 1e-3 of PyTorch against itself: the native path's
        #  re-associated FPN / Winograd layers are a larger perturbation than 1e-6 of the input, amplified by the same factor)
        assert attn[k] <= max(2e-4, 8 * attn_noise[k]), (k, attn[k], attn_noise[k])
    assert overall <= max(2e-3, 1.5 * overall_yard), (overall, overall_yard)
    assert worst_tight[0] <= 2e-3, worst_tight
    assert excess[0] <= max(2e-3, 8 * excess[1]), excess




def test_train_step_full_size_teacher_forced(shipped_cfg, checkpoint):
    """BASELINE config 4 at full size with the cascade's one discontinuity removed: both trees -- the native path and the
    oracle module tree on the GPU (plain PyTorch-ROCm) -- are given the SAME per-stage hypotheses (the oracle tree's own
    free-running ones), so an argmax flip in one tree cannot move the other's sampling planes, and winners carry no
    gradient (models/MVS4Net.py:78-111: depth is detached between stages).  What is left is a smooth function of the
    parameters, compared per parameter tensor.

    Yardstick per tensor: the PyTorch-ROCm step against ITSELF -- run twice on identical inputs (its atomics: grid_sample
    and weight-gradient backward) and once with the images perturbed by 1e-6 relative.  Measured (profiles/r05_*parity_model):
    even with the hypotheses pinned the step is ill-conditioned -- the fixture's sharpened prob heads saturate the softmax and
    the OT loss takes logs of it: PyTorch reproduces only 14 of the 167 gradient tensors to 1e-3 (median 1.1e-2, worst
    3.6e-2), so "every gradient to 1e-3" cannot be asked of ANY fp32 implementation here.  What is asserted: the loss to
    1e-5 relative (measured: equal to 7 digits); every tensor within max(2e-3, 8 x its own PyTorch yardstick) -- the same
    factor as for the attention volumes: the native path's re-associated layers deviate from PyTorch-ROCm like a ~5e-6
    relative input perturbation would (stage-4 attention 5.5x, the worst gradient tensor reg.3.conv0.bn.bias 3.9x the 1e-6
    yardstick; median tensor 5.7e-3 against a yardstick of 1.1e-2; the 14 well-conditioned tensors within 1.4e-3); the whole
    gradient vector within 1.5 x the yardstick's.  An O(1) error -- a wrong layer, a dropped term -- in any tensor whose
    yardstick is below ~10 % fails."""
    _train_step_full_size_teacher_forced(shipped_cfg, checkpoint, "", False)


def test_train_step_full_size_teacher_forced_well_conditioned(shipped_cfg, checkpoint):
    """The same teacher-forced full-size step with DEFAULT-INITIALISED prob heads (the fixture's sharpened heads are the
    stress case above): the softmax in front of the OT loss is not saturated, PyTorch-ROCm reproduces itself, and every
    parameter gradient of the native path is held to 1e-3 relative L2 against the oracle tree (models/MVS4Net.py:78-155)."""
    _train_step_full_size_teacher_forced(shipped_cfg, checkpoint, "_default_heads", True)


def test_eval_plans_follow_in_place_parameter_updates(shipped_cfg, checkpoint):
    """The folded-BatchNorm eval plans are stamped with every parameter's / buffer's version and storage address: an
    in-place update while the model stays in eval mode (EMA swap, ``copy_``, ``p.data = ...``) is picked up by the next
    forward instead of being served from stale packed weights."""
    imgs, proj, dv = to_dev(*make_inputs(nviews=3, H=64, W=128, seed=5))
    m = MVS4net(**shipped_cfg)
    m.load_state_dict(checkpoint, strict=True)
    m.to(DEV).eval()
    before = m(imgs, proj, dv)["stage4"]["attn_weight"].clone()
    plans_a = m._get_plans()
    assert m._get_plans()[0] is plans_a[0]                       # nothing changed: same plans
    with torch.no_grad():
        m.reg[3].conv0.conv.weight.mul_(1.5)                     # in place through the tensor
        m.feature.conv0[0].bn.running_mean.add_(0.05)            # a buffer
        m.reg[2].prob.weight.data = m.reg[2].prob.weight.data * 0.5          # storage replaced
    after = m(imgs, proj, dv)["stage4"]["attn_weight"]
    assert m._get_plans()[0] is not plans_a[0]
    fresh = MVS4net(**shipped_cfg)
    fresh.load_state_dict(m.state_dict(), strict=True)
    fresh.to(DEV).eval()
    want = fresh(imgs, proj, dv)["stage4"]["attn_weight"]
    assert torch.equal(after, want)
    assert not torch.equal(after, before)
