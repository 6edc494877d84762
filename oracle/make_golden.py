#!/usr/bin/env python3
"""Generate golden fixtures from the REFERENCE implementation.  TEST INFRASTRUCTURE.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

It imports the reference's ``models`` package read-only, runs it on seeded
synthetic inputs and stores *arrays only* (inputs + expected outputs) under
``tests/golden/*.npz`` -- fixtures G1..G7 of SURVEY.md section 8c.  No reference
source text is copied anywhere.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from mvster_amd.synthetic import make_inputs, randomize_state  # noqa: E402


def _import_reference():
    # the reference package is also called ``models``; make sure ours is not picked up
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import models.mvs4net_utils as U  # noqa
    import models.MVS4Net as M  # noqa
    sys.path.remove(REF)
    return U, M


def npz(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def composed(pm):
    """K @ [R|t] composition done with the reference's own two lines (stagenet)."""
    p = pm[:, 0].clone()
    p[:, :3, :4] = torch.matmul(pm[:, 1, :3, :3], pm[:, 0, :3, :4])
    return p


def g1_warp(U):
    torch.manual_seed(11)
    out = {}
    # case a: same size source/ref, DTU-like cameras
    _, proj, dv = make_inputs(nviews=3, H=256, W=320, seed=3)
    pm = proj["stage1"]  # 32x40
    ref_p, src_p = composed(pm[:, 0]), composed(pm[:, 1])
    depth = U.init_inverse_range(dv, 4, torch.device("cpu"), torch.float32, 32, 40)
    depth = depth * (1 + 0.01 * torch.rand_like(depth))
    fea = torch.randn(1, 8, 32, 40)
    out.update(a_fea=fea, a_src=src_p, a_ref=ref_p, a_depth=depth, a_out=U.homo_warping(fea, src_p, ref_p, depth))
    # case b: source size != reference size
    fea = torch.randn(1, 8, 24, 36)
    out.update(b_fea=fea, b_out=U.homo_warping(fea, src_p, ref_p, depth))
    # case c: mostly out-of-bounds projections (large baseline)
    src_far = src_p.clone()
    src_far[:, 0, 3] += 9000.0
    fea = torch.randn(1, 8, 32, 40)
    out.update(c_fea=fea, c_src=src_far, c_out=U.homo_warping(fea, src_far, ref_p, depth))
    # case d: z == 0 exactly (identity rotation, zero translation, depth 0 at some pixels)
    eye = torch.eye(4).unsqueeze(0)
    dz = depth.clone()
    dz[:, :, ::3, ::5] = 0.0
    out.update(d_depth=dz, d_out=U.homo_warping(fea, eye, eye.clone(), dz))
    # case e: B=2, different cameras per batch element
    _, proj2, dv2 = make_inputs(nviews=3, H=256, W=320, seed=5, batch=2)
    pm2 = proj2["stage1"]
    r2, s2 = composed(pm2[:, 0]), composed(pm2[:, 2])
    d2 = U.init_inverse_range(dv2, 4, torch.device("cpu"), torch.float32, 32, 40)
    f2 = torch.randn(2, 8, 32, 40)
    out.update(e_fea=f2, e_src=s2, e_ref=r2, e_depth=d2, e_out=U.homo_warping(f2, s2, r2, d2))
    npz("g1_warp", **out)


def g2_aggregate(U):
    """Parameter-free part of stagenet: capture the regnet input."""
    torch.manual_seed(12)
    out = {}

    class Grab(torch.nn.Module):
        def forward(self, x):
            self.x = x.clone()
            return x.sum(1)  # [B,D,H,W] so that the rest of stagenet runs

    cases = [
        ("gc_t2", dict(nviews=5, C=16, G=4, D=4, group_cor=True, attn_fuse_d=True, attn_temp=2)),
        ("gc_t1", dict(nviews=3, C=32, G=8, D=8, group_cor=True, attn_fuse_d=True, attn_temp=1)),
        ("sq_t2", dict(nviews=4, C=8, G=8, D=4, group_cor=False, attn_fuse_d=True, attn_temp=2)),
        ("gc_nofuse", dict(nviews=3, C=16, G=4, D=4, group_cor=True, attn_fuse_d=False, attn_temp=2)),
        ("gc_b2", dict(nviews=3, C=8, G=4, D=4, group_cor=True, attn_fuse_d=True, attn_temp=2, batch=2)),
    ]
    for name, c in cases:
        B = c.get("batch", 1)
        h, w = 24, 40
        _, proj, dv = make_inputs(nviews=c["nviews"], H=h * 8, W=w * 8, seed=21, batch=B)
        pm = proj["stage1"]
        feats = [0.5 * torch.randn(B, c["C"], h, w) for _ in range(c["nviews"])]
        hypo = U.init_inverse_range(dv, c["D"], torch.device("cpu"), torch.float32, h, w)
        hypo = hypo * (1 + 0.02 * torch.rand_like(hypo))
        sn = U.stagenet(inverse_depth=True, mono=False, attn_fuse_d=c["attn_fuse_d"], vis_ETA=False,
                        attn_temp=c["attn_temp"]).eval()
        grab = Grab()
        sn(feats, pm, depth_hypo=hypo, regnet=grab, stage_idx=0, group_cor=c["group_cor"],
           group_cor_dim=c["G"], split_itv=0.5)
        out[name + "_feats"] = torch.stack(feats)
        out[name + "_proj"] = pm
        out[name + "_hypo"] = hypo
        out[name + "_cor"] = grab.x
        out[name + "_cfg"] = np.array([c["nviews"], c["C"], c["G"], c["D"], int(c["group_cor"]),
                                       int(c["attn_fuse_d"]), c["attn_temp"]], dtype=np.float32)
    npz("g2_aggregate", **out)


def g3_reg(U):
    torch.manual_seed(13)
    out = {}
    for name, net, shape in [
        ("reg2d_g8", U.reg2d(input_channel=8, base_channel=8), (1, 8, 8, 16, 24)),
        ("reg2d_g4", U.reg2d(input_channel=4, base_channel=8), (2, 4, 4, 32, 16)),
        ("reg3d_ds3", U.reg3d(in_channels=8, base_channels=8, down_size=3), (1, 8, 8, 16, 24)),
        ("reg3d_ds2", U.reg3d(in_channels=4, base_channels=8, down_size=2), (1, 4, 4, 16, 16)),
    ]:
        sd = randomize_state(net.state_dict(), seed=31, prob_gain=1.0)
        net.load_state_dict(sd)
        net.eval()
        x = torch.randn(*shape)
        with torch.no_grad():
            y = net(x)
        out[name + "_x"] = x
        out[name + "_y"] = y
        for k, v in sd.items():
            out[name + "/" + k] = v
    npz("g3_reg", **out)


def g4_select(U):
    torch.manual_seed(14)
    out = {}
    sn = U.stagenet(inverse_depth=True, mono=False, attn_fuse_d=True, attn_temp=2).eval()

    class Feed(torch.nn.Module):
        def __init__(self, logits):
            super().__init__()
            self.logits = logits

        def forward(self, x):
            return self.logits

    for name, D, stage_idx, ties in [("d8_s0", 8, 0, False), ("d4_s2", 4, 2, False), ("d4_s3_ties", 4, 3, True),
                                     ("d8_s1_b2", 8, 1, False)]:
        B = 2 if name.endswith("b2") else 1
        h, w = 12, 20
        _, proj, dv = make_inputs(nviews=2, H=h * 8, W=w * 8, seed=41, batch=B)
        logits = 3.0 * torch.randn(B, D, h, w)
        if ties:
            logits[:, 1] = logits[:, 3]           # exact ties between bins 1 and 3
            logits[:, :, ::2, ::2] = 0.25         # all bins equal on a sub-lattice
        hypo = U.init_inverse_range(dv, D, torch.device("cpu"), torch.float32, h, w)
        hypo = hypo * (1 + 0.02 * torch.rand_like(hypo))
        feats = [torch.randn(B, 8, h, w) for _ in range(2)]
        with torch.no_grad():
            r = sn(feats, proj["stage1"], depth_hypo=hypo, regnet=Feed(logits), stage_idx=stage_idx, group_cor=True,
                   group_cor_dim=4, split_itv=0.5)
        out[name + "_logits"] = logits
        out[name + "_hypo"] = hypo
        out[name + "_stage_idx"] = np.array(stage_idx)
        for k in ("depth", "photometric_confidence", "attn_weight", "inverse_min_depth", "inverse_max_depth"):
            out[name + "_" + k] = r[k]
    npz("g4_select", **out)


def g5_sched(U):
    torch.manual_seed(15)
    out = {}
    dv = torch.tensor([[425.0, 933.8], [300.0, 1100.0]])
    out["dv"] = dv
    out["init_inverse_8"] = U.init_inverse_range(dv, 8, torch.device("cpu"), torch.float32, 6, 10)
    out["init_range_8"] = U.init_range(dv, 8, torch.device("cpu"), torch.float32, 6, 10)
    inv_min = 1.0 / (500 + 200 * torch.rand(2, 12, 20))
    inv_max = inv_min - 2e-4 * (0.5 + torch.rand(2, 12, 20))
    out["inv_min"], out["inv_max"] = inv_min, inv_max
    out["sched_inverse_8"] = U.schedule_inverse_range(inv_min, inv_max, 8, 24, 40)
    out["sched_inverse_4"] = U.schedule_inverse_range(inv_min, inv_max, 4, 24, 40)
    cur = 500 + 200 * torch.rand(2, 12, 20)
    itv = np.array([2.5, 3.0], dtype=np.float32)
    out["cur_depth"], out["itv"] = cur, itv
    out["sched_range_4"] = U.schedule_range(cur, 4, itv, 24, 40)
    npz("g5_sched", **out)


SHIPPED = dict(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
               stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True,
               group_cor_dim=[8, 8, 4, 4], inverse_depth=True, agg_type="ConvBnReLU3D", dcn=False, pos_enc=0,
               mono=True, asff=False, attn_temp=2, attn_fuse_d=True)


def g6_g7_end_to_end(U, M):
    torch.manual_seed(16)
    model = M.MVS4net(**SHIPPED)
    sd = randomize_state(model.state_dict(), seed=7, prob_gain=20.0)
    model.load_state_dict(sd)
    npz("g7_checkpoint", **{k: v for k, v in sd.items()})

    H, W, N = 128, 192, 5
    imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=1)
    cap = {}
    hooks = []
    for s in range(4):
        def pre(mod, args, s=s):
            cap["stage%d_cor_feats" % (s + 1)] = args[0].detach().clone()

        def post(mod, args, res, s=s):
            cap["stage%d_logits" % (s + 1)] = res.detach().clone()
        hooks.append(model.reg[s].register_forward_pre_hook(pre))
        hooks.append(model.reg[s].register_forward_hook(post))
    model.eval()
    with torch.no_grad():
        out = model(imgs, proj, dv)
    arrs = dict(H=np.array(H), W=np.array(W), N=np.array(N))
    for s in range(1, 5):
        st = out["stage%d" % s]
        for k, v in st.items():
            arrs["stage%d_%s" % (s, k)] = v
        aw = st["attn_weight"]
        top2 = aw.topk(2, dim=1)[0]
        arrs["stage%d_margin" % s] = top2[:, 0] - top2[:, 1]
    arrs.update(cap)
    # a few feeder (FPN4) outputs, to localise a mismatch
    with torch.no_grad():
        for v, s in ((1, 1), (0, 2), (2, 3), (0, 4)):
            arrs["feat_v%d_stage%d" % (v, s)] = model.feature(imgs[v])["stage%d" % s]
    npz("g6_eval", **arrs)

    # train-mode forward + loss + a few parameter gradients (B=2 so that BN batch stats are defined)
    for h in hooks:
        h.remove()
    Ht, Wt, Nt = 64, 64, 3
    imgs, proj, dv = make_inputs(nviews=Nt, H=Ht, W=Wt, seed=2, batch=2)
    model.train()
    for p in model.parameters():
        p.grad = None
    out = model(imgs, proj, dv)
    g = torch.Generator().manual_seed(5)
    depth_gt, mask = {}, {}
    for s in range(1, 5):
        hs, ws = Ht // 2 ** (4 - s), Wt // 2 ** (4 - s)
        depth_gt["stage%d" % s] = 500 + 300 * torch.rand(2, hs, ws, generator=g)
        mask["stage%d" % s] = (torch.rand(2, hs, ws, generator=g) > 0.2).float()
    loss, l1s, ots, rng = M.MVS4net_loss(out, depth_gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1],
                                         inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)
    loss.backward()
    arrs = dict(H=np.array(Ht), W=np.array(Wt), N=np.array(Nt), loss=loss.detach(),
                ot=torch.stack([o.detach() for o in ots]), l1=torch.stack([o.detach() for o in l1s]))
    for s in range(1, 5):
        arrs["depth_gt_stage%d" % s] = depth_gt["stage%d" % s]
        arrs["mask_stage%d" % s] = mask["stage%d" % s]
        st = out["stage%d" % s]
        arrs["stage%d_attn_weight" % s] = st["attn_weight"].detach()
        arrs["stage%d_hypo_depth" % s] = st["hypo_depth"].detach()
        arrs["stage%d_depth" % s] = st["depth"].detach()
        if "mono_depth" in st:
            arrs["stage%d_mono_depth" % s] = st["mono_depth"].detach()
    named = dict(model.named_parameters())
    for k in ("reg.0.prob.weight", "reg.3.conv0.conv.weight", "reg.1.conv6.conv.weight", "feature.out1.weight",
              "feature.out4.weight", "feature.conv0.0.conv.weight", "feature.inner2.bias"):
        arrs["grad/" + k] = named[k].grad.detach()
    # BN running stats after the train step (momentum update) for two layers
    arrs["bn/reg.0.conv0.bn.running_mean"] = model.state_dict()["reg.0.conv0.bn.running_mean"]
    arrs["bn/feature.conv0.0.bn.running_var"] = model.state_dict()["feature.conv0.0.bn.running_var"]
    npz("g6_train", **arrs)


def g8_loss(U):
    """sinkhorn on its own (section 8f)."""
    torch.manual_seed(18)
    B, D, h, w = 2, 4, 6, 10
    hypo = torch.sort(500 + 300 * torch.rand(B, D, h, w), dim=1, descending=True)[0]
    attn = torch.softmax(2 * torch.randn(B, D, h, w), 1)
    gt = 500 + 300 * torch.rand(B, h, w)
    mask = torch.rand(B, h, w) > 0.3
    T, loss = U.sinkhorn(gt, hypo, attn, mask, iters=10, eps=1, continuous=False)
    npz("g8_sinkhorn", hypo=hypo, attn=attn, gt=gt, mask=mask, T=T, loss=loss)


def g8b_loss_continuous(U):
    """sinkhorn with the continuous ground-truth offset (ot_continous=True, train_mvs4.py:64): D x (D+1) cost."""
    torch.manual_seed(19)
    out = {}
    for name, (B, D, h, w) in (("d4", (2, 4, 6, 10)), ("d8", (1, 8, 5, 7))):
        inv = 1.0 / 900 + (1.0 / 450 - 1.0 / 900) * torch.rand(B, 1, h, w)
        step = 2e-5 * (0.5 + torch.rand(B, 1, h, w))
        hypo = 1.0 / (inv + step * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1))      # index 0 = farthest
        attn = torch.softmax(2 * torch.randn(B, D, h, w), 1)
        # ground truth inside, at the edge of and outside the hypothesis range; invalid (masked) pixels hold depth 0
        pos = -1.5 + (D + 2.0) * torch.rand(B, h, w)
        gt = 1.0 / (inv[:, 0] + step[:, 0] * pos)
        mask = torch.rand(B, h, w) > 0.25
        gt[~mask] = 0.0
        for iters, eps in ((10, 1.0), (3, 0.5)):
            T, loss = U.sinkhorn(gt, hypo, attn, mask, iters=iters, eps=eps, continuous=True)
            out["%s_it%d_T" % (name, iters)] = T
            out["%s_it%d_loss" % (name, iters)] = loss
        out.update({name + "_hypo": hypo, name + "_attn": attn, name + "_gt": gt, name + "_mask": mask})
    npz("g8b_sinkhorn_continuous", **out)


def g9_losses(M):
    """MVS4net_loss (range ratios, linear / inverse, continuous OT) and Blend_loss (MVS4Net.py:113-206) on the stage
    outputs stored in g6_train."""
    z = np.load(os.path.join(OUT, "g6_train.npz"))
    inputs, gt, mask = {}, {}, {}
    for s in range(1, 5):
        st = {k: torch.from_numpy(z["stage%d_%s" % (s, k)]) for k in ("depth", "hypo_depth", "attn_weight")}
        if s > 1:
            st["mono_depth"] = torch.from_numpy(z["stage%d_mono_depth" % s])
        inputs["stage%d" % s] = st
        gt["stage%d" % s] = torch.from_numpy(z["depth_gt_stage%d" % s])
        mask["stage%d" % s] = torch.from_numpy(z["mask_stage%d" % s])
    out = {"depth_max": torch.tensor([935.0, 900.0]), "depth_min": torch.tensor([425.0, 430.0])}
    cases = {
        "inv": dict(stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, mono=True),
        "lin_l1": dict(stage_lw=[0.5, 1, 1.5, 2], l1ot_lw=[0.3, 0.7], inverse_depth=False, ot_iter=3, ot_eps=1, mono=True),
        "cont": dict(stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=True,
                     mono=False),
    }
    for name, kw in cases.items():
        total, l1s, ots, rng = M.MVS4net_loss(inputs, gt, mask, **kw)
        out["mvs4_%s_total" % name] = total
        out["mvs4_%s_l1" % name] = torch.stack(l1s)
        out["mvs4_%s_ot" % name] = torch.stack(ots)
        out["mvs4_%s_range" % name] = torch.stack(rng)
        r = M.Blend_loss(inputs, gt, mask, depth_max=out["depth_max"], depth_min=out["depth_min"], **kw)
        out["blend_%s_total" % name] = r[0]
        out["blend_%s_l1" % name] = torch.stack(r[1])
        out["blend_%s_ot" % name] = torch.stack(r[2])
        out["blend_%s_range" % name] = torch.stack(r[3])
        out["blend_%s_epe_err3_err1" % name] = torch.stack([r[4], r[5], r[6]])
    npz("g9_losses", **out)


def _reference_functions(path, names, env):
    """Compile, at fixture-generation time, the named top-level functions / class methods of a reference script that
    cannot be imported whole here (test_mvs4.py parses the command line and imports cv2 / plyfile at load time).  The
    source is read from the read-only tree, compiled and executed in memory; nothing of it is written anywhere."""
    import ast
    tree = ast.parse(open(path).read(), path)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = dict(env)
            exec(compile(mod, path, "exec"), ns)
            found[node.name] = ns[node.name]
    missing = set(names) - set(found)
    assert not missing, missing
    return found


def g10_formats():
    """On-disk formats either side of the path, from the reference's own readers / writers:
    datasets/data_io.py (imported under an empty stand-in ``cv2`` module: read_pfm / save_pfm never touch it),
    test_mvs4.py:94-155 (read_camera_parameters, read_pair_file, write_cam: extracted with ast, the script itself parses
    argv on import) and datasets/general_eval4.py (MVSDataset run end to end on a synthetic scan directory whose images
    already have an admissible size, so its only cv2 call, ``cv2.resize(img, same size)``, is the identity)."""
    import io
    import tempfile
    import types
    from PIL import Image
    cv2 = types.ModuleType("cv2")

    def _resize_same_size_only(img, size):
        assert (img.shape[1], img.shape[0]) == tuple(size), "fixture must not need a real resize"
        return img
    cv2.resize = _resize_same_size_only
    had = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2
    sys.path.insert(0, REF)
    try:
        import datasets.data_io as DIO
        import datasets.general_eval4 as GE
    finally:
        sys.path.remove(REF)
        if had is None:
            del sys.modules["cv2"]
    T = _reference_functions(os.path.join(REF, "test_mvs4.py"),
                             ("read_camera_parameters", "read_pair_file", "write_cam"), {"np": np})
    rs = np.random.RandomState(10)
    out = {}
    tmp = tempfile.mkdtemp(prefix="g10_")

    def raw(path):
        return np.frombuffer(open(path, "rb").read(), dtype=np.uint8)

    # --- PFM: grey, H x W x 1, colour, big-endian data, a non-unit scale; the bytes the reference writes and what it reads
    pfm_cases = {
        "grey": (rs.rand(5, 7).astype(np.float32) * 900, 1),
        "grey1": (rs.randn(4, 6, 1).astype(np.float32), 1),
        "color": (rs.rand(4, 6, 3).astype(np.float32), 2.5),
        "big": (rs.randn(3, 5).astype(np.float32).astype(">f4"), 1),
        "special": (np.array([[0.0, -0.0, np.inf], [-np.inf, np.nan, 1e-45]], dtype=np.float32), 1),
    }
    for name, (img, scale) in pfm_cases.items():
        path = os.path.join(tmp, name + ".pfm")
        DIO.save_pfm(path, img, scale)
        back, sc = DIO.read_pfm(path)
        out["pfm_%s_image" % name] = img.astype(np.float32)
        out["pfm_%s_image_big_endian" % name] = np.array(img.dtype.byteorder == ">")
        out["pfm_%s_scale_in" % name] = np.array(scale, dtype=np.float64)
        out["pfm_%s_bytes" % name] = raw(path)
        out["pfm_%s_read" % name] = np.ascontiguousarray(back).astype(np.float32)
        out["pfm_%s_read_shape" % name] = np.array(back.shape)
        out["pfm_%s_scale_out" % name] = np.array(sc, dtype=np.float64)
    # a hand-built big-endian file with a "1.0" scale line (what other MVSNet tools write), read by the reference
    vals = np.array([[1.5, -2.0], [3.25, 4.0], [5.0, 6.5]], dtype=">f4")
    path = os.path.join(tmp, "hand.pfm")
    open(path, "wb").write(b"Pf\n2 3\n1.0\n" + vals.tobytes())
    back, sc = DIO.read_pfm(path)
    out["pfm_hand_bytes"] = raw(path)
    out["pfm_hand_read"] = np.ascontiguousarray(back).astype(np.float32)
    out["pfm_hand_scale_out"] = np.array(sc, dtype=np.float64)
    for bad, blob in (("magic", b"P6\n2 2\n1.0\n"), ("dims", b"Pf\n2x2\n1.0\n")):
        path = os.path.join(tmp, bad + ".pfm")
        open(path, "wb").write(blob)
        try:
            DIO.read_pfm(path)
            msg = ""
        except Exception as e:          # noqa: BLE001
            msg = str(e)
        out["pfm_bad_%s_bytes" % bad] = raw(path)
        out["pfm_bad_%s_message" % bad] = np.array(msg)
    for bad, img in (("dtype", np.zeros((2, 2), np.float64)), ("shape", np.zeros((2, 2, 2), np.float32))):
        try:
            DIO.save_pfm(os.path.join(tmp, "x.pfm"), img)
            msg = ""
        except Exception as e:          # noqa: BLE001
            msg = str(e)
        out["pfm_bad_%s_message" % bad] = np.array(msg)

    # --- a synthetic scan: five views of 64 x 128, DTU-like cameras; cam files written by the reference's write_cam
    scan = "scan_g10"
    os.makedirs(os.path.join(tmp, scan, "cams"))
    os.makedirs(os.path.join(tmp, scan, "images"))
    H, W, nv = 64, 128, 5
    for v in range(nv):
        cam = np.zeros((2, 4, 4), dtype=np.float32)
        ang = 0.05 * v
        cam[0] = np.eye(4)
        cam[0, :3, :3] = [[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]
        cam[0, :3, 3] = [-60.0 * v + 0.123456789, 3.5 * v, 0.25 * v]
        cam[1, :3, :3] = [[2892.33 * W / 1600 * 4, 0, W / 2 * 4 + 0.7], [0, 2883.18 * H / 1200 * 4, H / 2 * 4 - 0.3], [0, 0, 1]]
        cam[1, 3] = [425.0 + v, 2.5, 192, 425.0 + v + 2.5 * 192] if v != 2 else [425.0, 1.25, 0, 0]
        path = os.path.join(tmp, scan, "cams", "%08d_cam.txt" % v)
        T["write_cam"](path, cam)
        out["cam%d_array" % v] = cam
        out["cam%d_text" % v] = raw(path)
        K, E = T["read_camera_parameters"](path)
        out["cam%d_intrinsics" % v] = K
        out["cam%d_extrinsics" % v] = E
        yy, xx = np.mgrid[0:H, 0:W]
        img = np.stack([(xx * 2 + 17 * v) % 256, (yy * 3 + xx) % 256, (yy * xx // 7 + 40 * v) % 256], -1).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="JPEG", quality=95)
        open(os.path.join(tmp, scan, "images", "%08d.jpg" % v), "wb").write(buf.getvalue())
        out["img%d_jpeg" % v] = np.frombuffer(buf.getvalue(), dtype=np.uint8)
    # the two-field depth line (no plane count) some T&T cam files carry
    path2 = os.path.join(tmp, "two_field_cam.txt")
    txt = open(os.path.join(tmp, scan, "cams", "00000001_cam.txt")).read().rstrip("\n").rsplit("\n", 1)[0] + "\n0.5 0.0125\n"
    open(path2, "w").write(txt)
    out["cam_two_field_text"] = raw(path2)
    pair_txt = "5\n0\n3 1 0.9 2 0.8 3 0.1\n1\n2 0 1.5 2 0.25\n2\n0\n3\n1 0 1.0\n4\n10 3 2.0 2 1.9 1 1.8 0 1.7 3 1.6 3 1.5 3 1.4 3 1.3 3 1.2 3 1.1\n"
    open(os.path.join(tmp, scan, "pair.txt"), "w").write(pair_txt)
    out["pair_text"] = np.frombuffer(pair_txt.encode(), dtype=np.uint8)
    pairs = T["read_pair_file"](os.path.join(tmp, scan, "pair.txt"))
    out["pair_refs"] = np.array([r for r, _ in pairs])
    out["pair_src_counts"] = np.array([len(s) for _, s in pairs])
    out["pair_srcs_flat"] = np.array([x for _, s in pairs for x in s])

    # --- the reference's evaluation dataset on that scan (general_eval4.py:8-188): metas, per-sample tensors
    for tag, nviews, interval_scale in (("n3", 3, 1.06), ("n5", 5, {scan: 0.8})):
        ds = GE.MVSDataset(tmp, [scan], "test", nviews, interval_scale=interval_scale, max_h=H, max_w=W, fix_res=False)
        out["ds_%s_len" % tag] = np.array(len(ds))
        out["ds_%s_meta_ref" % tag] = np.array([m[1] for m in ds.metas])
        out["ds_%s_meta_src_counts" % tag] = np.array([len(m[2]) for m in ds.metas])
        out["ds_%s_meta_srcs_flat" % tag] = np.array([x for m in ds.metas for x in m[2]])
        for i in range(len(ds)):
            smp = ds[i]
            if i == 0:
                out["ds_%s_%d_imgs" % (tag, i)] = np.stack(smp["imgs"])
            for st in ("stage1", "stage2", "stage3", "stage4"):
                out["ds_%s_%d_%s" % (tag, i, st)] = smp["proj_matrices"][st]
            out["ds_%s_%d_depth_values" % (tag, i)] = smp["depth_values"]
            out["ds_%s_%d_filename" % (tag, i)] = np.array(smp["filename"])
    K4, E4, dmin, itv = ds.read_cam_file(path2, 1.06)
    out["cam_two_field_read"] = np.array([dmin, itv], dtype=np.float64)
    out["cam_two_field_intrinsics"] = K4
    out["ds_dims"] = np.array([H, W, nv])
    npz("g10_formats", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    U, M = _import_reference()
    only = set(sys.argv[1:])          # e.g. ``make_golden.py g8b g9`` adds fixtures without rewriting the others

    def want(tag):
        return not only or tag in only
    if want("g1"):
        g1_warp(U)
    if want("g2"):
        g2_aggregate(U)
    if want("g3"):
        g3_reg(U)
    if want("g4"):
        g4_select(U)
    if want("g5"):
        g5_sched(U)
    if want("g6"):
        g6_g7_end_to_end(U, M)
    if want("g8"):
        g8_loss(U)
    if want("g8b"):
        g8b_loss_continuous(U)
    if want("g9"):
        g9_losses(M)
    if want("g10"):
        g10_formats()


if __name__ == "__main__":
    main()
