"""CPU oracle for the MVSTER cost-volume hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm
for the path SURVEY.md section 8 scopes.  It exists so that the HIP path in
``mvster_amd`` can be checked against something that was itself pinned to the
reference: ``oracle/make_golden.py`` imports the reference (read-only, in the
build container) and commits its outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against them.

Parity status: PINNED (golden vectors generated from the reference's own code
running on torch 2.10 CPU; the reference ships no tests of its own).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Nothing under ``mvster_amd/`` imports it.

Every function cites the reference lines (``/root/reference/...``) it follows.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
# depth-hypothesis schedulers                                                  #
# --------------------------------------------------------------------------- #
def init_inverse_range(depth_values, ndepths, H, W):
    """Stage-1 hypotheses, uniform in 1/depth, index 0 = farthest.
    Follows models/mvs4net_utils.py:71-77."""
    inv_near = 1.0 / depth_values[:, 0]
    inv_far = 1.0 / depth_values[:, -1]
    ramp = torch.arange(0, ndepths, dtype=depth_values.dtype, device=depth_values.device)
    ramp = ramp.view(1, -1, 1, 1).repeat(1, 1, H, W) / (ndepths - 1)
    inv = inv_far[:, None, None, None] + (inv_near - inv_far)[:, None, None, None] * ramp
    return 1.0 / inv


def schedule_inverse_range(inverse_min_depth, inverse_max_depth, ndepths, H, W):
    """Stage>1 hypotheses: lerp between the previous stage's inverse bounds at
    half resolution, trilinear (align_corners) upsample to HxW, reciprocal.
    Follows models/mvs4net_utils.py:79-86."""
    ramp = torch.arange(0, ndepths, dtype=inverse_min_depth.dtype, device=inverse_min_depth.device)
    ramp = ramp.view(1, -1, 1, 1).repeat(1, 1, H // 2, W // 2) / (ndepths - 1)
    inv = inverse_max_depth[:, None] + (inverse_min_depth - inverse_max_depth)[:, None] * ramp
    inv = F.interpolate(inv.unsqueeze(1), [ndepths, H, W], mode="trilinear", align_corners=True).squeeze(1)
    return 1.0 / inv


def init_range(depth_values, ndepths, H, W):
    """Linear-depth stage-1 hypotheses.  Follows models/mvs4net_utils.py:61-69."""
    dmin = depth_values[:, 0]
    dmax = depth_values[:, -1]
    step = ((dmax - dmin) / (ndepths - 1))[:, None, None]
    idx = torch.arange(0, ndepths, dtype=depth_values.dtype, device=depth_values.device).reshape(1, -1)
    samples = dmin.unsqueeze(1) + idx * step.squeeze(1)
    return samples.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, H, W)


def schedule_range(cur_depth, ndepth, depth_interval_pixel, H, W):
    """Linear-depth stage>1 hypotheses.  Follows models/mvs4net_utils.py:88-99.
    ``depth_interval_pixel`` is a [B] tensor here (the reference passes a numpy
    array, which only works for CPU tensors -- SURVEY.md section 5)."""
    itv = torch.as_tensor(depth_interval_pixel, dtype=cur_depth.dtype, device=cur_depth.device)
    lo = cur_depth - ndepth / 2 * itv[:, None, None]
    hi = cur_depth + ndepth / 2 * itv[:, None, None]
    step = (hi - lo) / (ndepth - 1)
    idx = torch.arange(0, ndepth, dtype=cur_depth.dtype, device=cur_depth.device).reshape(1, -1, 1, 1)
    samples = lo.unsqueeze(1) + idx * step.unsqueeze(1)
    return F.interpolate(samples.unsqueeze(1), [ndepth, H, W], mode="trilinear", align_corners=True).squeeze(1)


# --------------------------------------------------------------------------- #
# projection + warp                                                            #
# --------------------------------------------------------------------------- #
def compose_projection(pm):
    """[B,2,4,4] (extrinsic, intrinsic) -> [B,4,4] with rows 0..2 = K @ E[:3,:4]
    and row 3 kept from the extrinsic.  Follows models/mvs4net_utils.py:1032-1035."""
    out = pm[:, 0].clone()
    out[:, :3, :4] = torch.matmul(pm[:, 1, :3, :3], pm[:, 0, :3, :4])
    return out


def warp_grid(src_proj, ref_proj, depth_values, Hs, Ws, origin=(0, 0)):
    """Normalised sampling grid [B, D, Hr*Wr, 2] of a source view for every
    reference pixel/hypothesis.  Follows models/mvs4net_utils.py:23-45.
    ``origin`` = (y0, x0): ``depth_values`` is the [y0:y0+Hr, x0:x0+Wr] window of a larger reference map (same
    arithmetic per pixel as the full map; used to check full-size outputs on a window in seconds)."""
    B, D, Hr, Wr = depth_values.shape
    dev = depth_values.device
    proj = torch.matmul(src_proj, torch.inverse(ref_proj))
    rot = proj[:, :3, :3]
    trans = proj[:, :3, 3:4]
    yy, xx = torch.meshgrid(torch.arange(origin[0], origin[0] + Hr, dtype=torch.float32, device=dev),
                            torch.arange(origin[1], origin[1] + Wr, dtype=torch.float32, device=dev), indexing="ij")
    yy = yy.reshape(Hr * Wr)
    xx = xx.reshape(Hr * Wr)
    pix = torch.stack((xx, yy, torch.ones_like(xx))).unsqueeze(0).repeat(B, 1, 1)
    rot_pix = torch.matmul(rot, pix)
    cam = rot_pix.unsqueeze(2).repeat(1, 1, D, 1) * depth_values.reshape(B, 1, D, -1)
    cam = cam + trans.reshape(B, 3, 1, 1)
    z = cam[:, 2:3]
    z[z == 0] = 1e-9
    xy = cam[:, :2] / z
    gx = xy[:, 0] / ((Ws - 1) / 2) - 1
    gy = xy[:, 1] / ((Hs - 1) / 2) - 1
    return torch.stack((gx, gy), dim=3)


def homo_warping(src_fea, src_proj, ref_proj, depth_values, origin=(0, 0)):
    """[B,C,Hs,Ws] source feature -> [B,C,D,Hr,Wr] warped volume (bilinear,
    zeros padding, align_corners=True).  Follows models/mvs4net_utils.py:13-59."""
    B, C, Hs, Ws = src_fea.shape
    _, D, Hr, Wr = depth_values.shape
    with torch.no_grad():
        grid = warp_grid(src_proj, ref_proj, depth_values, Hs, Ws, origin)
    out = F.grid_sample(src_fea, grid.reshape(B, D * Hr, Wr, 2), mode="bilinear",
                        padding_mode="zeros", align_corners=True)
    return out.reshape(B, C, D, Hr, Wr)


# --------------------------------------------------------------------------- #
# correlation + epipolar attention aggregation                                 #
# --------------------------------------------------------------------------- #
def aggregate_views(features, proj_matrices, depth_hypo, group_cor, group_cor_dim,
                    attn_temp=2.0, attn_fuse_d=True, origin=(0, 0)):
    """Per-stage parameter-free part: warp every source view, correlate with the
    reference feature, weight each view by a softmax over the depth axis, sum
    over views and normalise.  Returns cor_feats [B,G|C,D,h,w].
    Follows models/mvs4net_utils.py:1015-1060.  With ``origin`` = (y0, x0) the reference feature and
    ``depth_hypo`` are the window [y0:y0+h, x0:x0+w] of a larger reference map (source features stay whole)."""
    pms = torch.unbind(proj_matrices, 1)
    ref_fea, src_feas = features[0], features[1:]
    B, D, H, W = depth_hypo.shape
    C = ref_fea.shape[1]
    ref_vol = ref_fea.unsqueeze(2).repeat(1, 1, D, 1, 1)
    weight_sum = 1e-8
    acc = 0
    ref_p = compose_projection(pms[0])
    for src_fea, pm in zip(src_feas, pms[1:]):
        src_p = compose_projection(pm)
        warped = homo_warping(src_fea, src_p, ref_p, depth_hypo, origin)
        if group_cor:
            G = group_cor_dim
            cor = (warped.reshape(B, G, C // G, D, H, W) * ref_vol.reshape(B, G, C // G, D, H, W)).mean(2)
        else:
            cor = (ref_vol - warped) ** 2
        if attn_fuse_d:
            wgt = torch.softmax(cor.sum(1) / attn_temp, 1) / math.sqrt(C)   # [B,D,h,w]
            weight_sum = weight_sum + wgt
            acc = acc + wgt.unsqueeze(1) * cor
        else:
            wgt = torch.softmax(cor.sum(1), 1).max(1)[0]                      # [B,h,w]
            weight_sum = weight_sum + wgt
            acc = acc + wgt.unsqueeze(1).unsqueeze(1) * cor
    if attn_fuse_d:
        return acc / weight_sum.unsqueeze(1)
    return acc / weight_sum.unsqueeze(1).unsqueeze(1)


# --------------------------------------------------------------------------- #
# depth selection                                                              #
# --------------------------------------------------------------------------- #
def select_depth(logits, depth_hypo, stage_idx, inverse_depth, split_itv, training=False):
    """softmax over D, winner-take-all depth, confidence, inverse-range outputs.
    Follows models/mvs4net_utils.py:1068-1088."""
    attn = F.softmax(logits, dim=1)
    idx = attn.max(1, keepdim=True)[1]
    depth = torch.gather(depth_hypo, 1, idx).squeeze(1)
    if not training:
        with torch.no_grad():
            conf = attn.max(1)[0]
            conf = F.interpolate(conf.unsqueeze(1), scale_factor=2 ** (3 - stage_idx), mode="bilinear",
                                 align_corners=True).squeeze(1)
    else:
        conf = torch.tensor(0.0, dtype=torch.float32, device=logits.device)
    out = {"depth": depth, "photometric_confidence": conf, "hypo_depth": depth_hypo, "attn_weight": attn}
    if inverse_depth:
        itv = 1.0 / depth_hypo[:, 2] - 1.0 / depth_hypo[:, 1]
        out["inverse_min_depth"] = 1 / depth + split_itv * itv
        out["inverse_max_depth"] = 1 / depth - split_itv * itv
    return out


# --------------------------------------------------------------------------- #
# parameterised blocks (state_dict names identical to the reference)           #
# --------------------------------------------------------------------------- #
class _CBR3d(nn.Module):
    """Conv3d(no bias) + BatchNorm3d + ReLU.  models/mvs4net_utils.py:116-123."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


def _up3d(cin, cout, k, p, op, s):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=k, padding=p, output_padding=op, stride=s, bias=False),
                         nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


class Reg2d(nn.Module):
    """Cost regularisation U-Net, (1,3,3) strided layers + 3x3x3 bottlenecks.
    models/mvs4net_utils.py:870-912."""

    def __init__(self, input_channel=128, base_channel=32):
        super().__init__()
        c = base_channel
        k, p = (1, 3, 3), (0, 1, 1)
        self.conv0 = _CBR3d(input_channel, c, kernel_size=k, pad=p)
        self.conv1 = _CBR3d(c, c * 2, kernel_size=k, stride=(1, 2, 2), pad=p)
        self.conv2 = _CBR3d(c * 2, c * 2)
        self.conv3 = _CBR3d(c * 2, c * 4, kernel_size=k, stride=(1, 2, 2), pad=p)
        self.conv4 = _CBR3d(c * 4, c * 4)
        self.conv5 = _CBR3d(c * 4, c * 8, kernel_size=k, stride=(1, 2, 2), pad=p)
        self.conv6 = _CBR3d(c * 8, c * 8)
        self.conv7 = _up3d(c * 8, c * 4, k, p, (0, 1, 1), (1, 2, 2))
        self.conv9 = _up3d(c * 4, c * 2, k, p, (0, 1, 1), (1, 2, 2))
        self.conv11 = _up3d(c * 2, c, k, p, (0, 1, 1), (1, 2, 2))
        self.prob = nn.Conv3d(8, 1, 1, stride=1, padding=0)

    def forward(self, x):
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        x = self.conv6(self.conv5(c4))
        x = c4 + self.conv7(x)
        x = c2 + self.conv9(x)
        x = c0 + self.conv11(x)
        return self.prob(x).squeeze(1)


class Reg3d(nn.Module):
    """Full 3x3x3 U-Net variant.  models/mvs4net_utils.py:914-965."""

    def __init__(self, in_channels, base_channels, down_size=3):
        super().__init__()
        c = base_channels
        self.down_size = down_size
        self.conv0 = _CBR3d(in_channels, c)
        self.conv1 = _CBR3d(c, c * 2, stride=2)
        self.conv2 = _CBR3d(c * 2, c * 2)
        if down_size >= 2:
            self.conv3 = _CBR3d(c * 2, c * 4, stride=2)
            self.conv4 = _CBR3d(c * 4, c * 4)
        if down_size >= 3:
            self.conv5 = _CBR3d(c * 4, c * 8, stride=2)
            self.conv6 = _CBR3d(c * 8, c * 8)
            self.conv7 = _up3d(c * 8, c * 4, 3, 1, 1, 2)
        if down_size >= 2:
            self.conv9 = _up3d(c * 4, c * 2, 3, 1, 1, 2)
        self.conv11 = _up3d(c * 2, c, 3, 1, 1, 2)
        self.prob = nn.Conv3d(c, 1, 3, stride=1, padding=1, bias=False)

    def forward(self, x):
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        if self.down_size == 3:
            c4 = self.conv4(self.conv3(c2))
            x = self.conv6(self.conv5(c4))
            x = c4 + self.conv7(x)
            x = c2 + self.conv9(x)
        elif self.down_size == 2:
            x = self.conv4(self.conv3(c2))
            x = c2 + self.conv9(x)
        else:
            x = c2
        x = c0 + self.conv11(x)
        return self.prob(x).squeeze(1)


class _CBR2d(nn.Module):
    """Conv2d(no bias) + BatchNorm2d + optional ReLU.  models/mvs4net_utils.py:224-246."""

    def __init__(self, cin, cout, k, stride=1, relu=True, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, momentum=0.1)
        self.relu = relu

    def forward(self, x):
        x = self.bn(self.conv(x))
        return F.relu(x, inplace=True) if self.relu else x


class FPN4(nn.Module):
    """4-scale feature pyramid, out channels 8c/4c/2c/c at 1/8,1/4,1/2,1.
    models/mvs4net_utils.py:419-502 (gn=False, dcn=False)."""

    def __init__(self, base_channels=8):
        super().__init__()
        c = base_channels
        self.conv0 = nn.Sequential(_CBR2d(3, c, 3, 1, padding=1), _CBR2d(c, c, 3, 1, padding=1))
        self.conv1 = nn.Sequential(_CBR2d(c, 2 * c, 5, stride=2, padding=2), _CBR2d(2 * c, 2 * c, 3, 1, padding=1),
                                   _CBR2d(2 * c, 2 * c, 3, 1, padding=1))
        self.conv2 = nn.Sequential(_CBR2d(2 * c, 4 * c, 5, stride=2, padding=2), _CBR2d(4 * c, 4 * c, 3, 1, padding=1),
                                   _CBR2d(4 * c, 4 * c, 3, 1, padding=1))
        self.conv3 = nn.Sequential(_CBR2d(4 * c, 8 * c, 5, stride=2, padding=2), _CBR2d(8 * c, 8 * c, 3, 1, padding=1),
                                   _CBR2d(8 * c, 8 * c, 3, 1, padding=1))
        f = 8 * c
        self.inner1 = nn.Conv2d(4 * c, f, 1, bias=True)
        self.inner2 = nn.Conv2d(2 * c, f, 1, bias=True)
        self.inner3 = nn.Conv2d(c, f, 1, bias=True)
        self.out1 = nn.Conv2d(f, 8 * c, 1, bias=False)
        self.out2 = nn.Conv2d(f, 4 * c, 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(f, 2 * c, 3, padding=1, bias=False)
        self.out4 = nn.Conv2d(f, c, 3, padding=1, bias=False)
        self.out_channels = [8 * c, 4 * c, 2 * c, c]

    def forward(self, x):
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
        f = c3
        o1 = self.out1(f)
        f = up(f) + self.inner1(c2)
        o2 = self.out2(f)
        f = up(f) + self.inner2(c1)
        o3 = self.out3(f)
        f = up(f) + self.inner3(c0)
        o4 = self.out4(f)
        return {"stage1": o1, "stage2": o2, "stage3": o3, "stage4": o4}

    def forward_window(self, x_win, origin, full_size):
        """The pyramid of a WINDOW of a large image: ``x_win`` = image[:, :, y0:y0+wh, x0:x0+ww] with ``origin`` = (y0, x0)
        and the window size multiples of 8, ``full_size`` = (H, W) of the whole image.  Same layers as ``forward``; the
        only step of FPN4 that is not translation-invariant -- the x2 bilinear up-sampling with align_corners=True
        (models/mvs4net_utils.py:485-495), whose source coordinate is dst * (n_in - 1) / (n_out - 1) of the WHOLE map --
        is evaluated in whole-map coordinates (``_up2_window``).  Window pixels closer than ~64 pixels to a window edge
        that is not an image edge see the window's zero padding instead of the image: compare the interior only.
        Lets tests check the full-size pyramid (minutes on the CPU as a whole) window by window in seconds;
        tests/test_oracle_golden.py checks it against ``forward`` itself."""
        (y0, x0), (H, W) = origin, full_size
        if y0 % 8 or x0 % 8 or x_win.shape[2] % 8 or x_win.shape[3] % 8:
            raise ValueError("forward_window: origin and size must be multiples of 8")
        c0 = self.conv0(x_win)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        f = c3
        o1 = self.out1(f)
        f = _up2_window(f, (y0 // 8, x0 // 8), (H // 8, W // 8)) + self.inner1(c2)
        o2 = self.out2(f)
        f = _up2_window(f, (y0 // 4, x0 // 4), (H // 4, W // 4)) + self.inner2(c1)
        o3 = self.out3(f)
        f = _up2_window(f, (y0 // 2, x0 // 2), (H // 2, W // 2)) + self.inner3(c0)
        o4 = self.out4(f)
        return {"stage1": o1, "stage2": o2, "stage3": o3, "stage4": o4}


def _up2_window(t, origin, full):
    """x2 bilinear up-sampling, align_corners=True, of the window ``t`` [B,C,n_y,n_x] (origin ``origin``) of a map of size
    ``full``, in the coordinates of the whole map: output sample Y of 2*full reads source Y * (full-1)/(2*full-1) (fp32,
    as ATen's area_pixel_compute_scale / compute_source_index_and_lambda do), taps i0 = floor and min(i0+1, full-1).  Taps
    that fall outside the window are clamped into it (window-edge samples only: outside the interior a caller compares)."""
    def axis(n0, n, size):
        Y = torch.arange(2 * n0, 2 * (n0 + n), dtype=torch.float32)
        scale = torch.tensor(float(size - 1), dtype=torch.float32) / torch.tensor(float(2 * size - 1), dtype=torch.float32)
        src = scale * Y
        i0 = src.floor().long()
        lam = src - i0.float()
        i1 = torch.clamp(i0 + 1, max=size - 1)
        return (i0 - n0).clamp(0, n - 1), (i1 - n0).clamp(0, n - 1), lam
    y0i, y1i, ly = axis(origin[0], t.shape[2], full[0])
    x0i, x1i, lx = axis(origin[1], t.shape[3], full[1])
    ly, lx = ly.view(1, 1, -1, 1).to(t.device), lx.view(1, 1, 1, -1).to(t.device)
    top, bot = t[:, :, y0i], t[:, :, y1i]
    return ((1 - ly) * ((1 - lx) * top[:, :, :, x0i] + lx * top[:, :, :, x1i]) +
            ly * ((1 - lx) * bot[:, :, :, x0i] + lx * bot[:, :, :, x1i]))


class MonoDepthDecoder(nn.Module):
    """Training-only auxiliary head.  models/mvs4net_utils.py:833-868."""

    def __init__(self):
        super().__init__()
        self.convblocks = nn.ModuleList([_CBR2d(64, 32, 3, 1, padding=1), _CBR2d(32, 16, 3, 1, padding=1),
                                         _CBR2d(16, 8, 3, 1, padding=1)])
        self.conv3x3 = nn.ModuleList([nn.Conv2d(64, 1, 3, 1, 1), nn.Conv2d(32, 1, 3, 1, 1), nn.Conv2d(16, 1, 3, 1, 1)])

    def forward(self, outputs, d_min, d_max):
        for i in range(1, 4):
            small = outputs["stage%d" % i]["mono_feat"]
            large = outputs["stage%d" % (i + 1)]["mono_feat"]
            small = F.interpolate(self.convblocks[i - 1](small), scale_factor=2, mode="nearest")
            disp = torch.sigmoid(self.conv3x3[i - 1](torch.cat([small, large], 1)))
            lo = (1 / d_max)[:, None, None, None]
            hi = (1 / d_min)[:, None, None, None]
            outputs["stage%d" % (i + 1)]["mono_depth"] = (1 / (lo + (hi - lo) * disp)).squeeze(1)
        return outputs


# --------------------------------------------------------------------------- #
# cascade driver                                                               #
# --------------------------------------------------------------------------- #
class OracleMVS4net(nn.Module):
    """CPU restatement of ``MVS4net`` (models/MVS4Net.py:9-111) for the shipped
    options (fpn, reg2d|reg3d, group_cor, inverse_depth, mono, attn_temp,
    attn_fuse_d).  Same constructor keywords, forward signature, output dict and
    state_dict keys as the reference."""

    def __init__(self, arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
                 stage_splits=(8, 8, 4, 4), depth_interals_ratio=(0.5, 0.5, 0.5, 1), group_cor=False,
                 group_cor_dim=(8, 8, 8, 8), inverse_depth=False, agg_type="ConvBnReLU3D", dcn=False, pos_enc=0,
                 mono=False, asff=False, attn_temp=2, attn_fuse_d=True, vis_ETA=False, vis_mono=False):
        super().__init__()
        assert arch_mode == "fpn" and agg_type == "ConvBnReLU3D" and not dcn and not asff and pos_enc == 0
        self.num_stage = num_stage
        self.stage_splits = list(stage_splits)
        self.depth_interals_ratio = list(depth_interals_ratio)
        self.group_cor = group_cor
        self.group_cor_dim = list(group_cor_dim)
        self.inverse_depth = inverse_depth
        self.mono = mono
        self.attn_temp = attn_temp
        self.attn_fuse_d = attn_fuse_d
        self.feature = FPN4(base_channels=fpn_base_channel)
        if mono:
            self.mono_depth_decoder = MonoDepthDecoder()
        self.reg = nn.ModuleList()
        down = [3, 3, 2, 2]
        for s in range(num_stage):
            cin = self.group_cor_dim[s] if group_cor else self.feature.out_channels[s]
            if reg_net == "reg2d":
                self.reg.append(Reg2d(input_channel=cin, base_channel=reg_channel))
            else:
                self.reg.append(Reg3d(in_channels=cin, base_channels=reg_channel, down_size=down[s]))

    def run_stage(self, feats, pm, depth_hypo, s, capture=None):
        cor = aggregate_views(feats, pm, depth_hypo, self.group_cor, self.group_cor_dim[s],
                              attn_temp=self.attn_temp, attn_fuse_d=self.attn_fuse_d)
        logits = self.reg[s](cor)
        if capture is not None:
            capture["cor_feats"] = cor
            capture["logits"] = logits
        out = select_depth(logits, depth_hypo, s, self.inverse_depth, self.depth_interals_ratio[s], self.training)
        if self.mono:
            out["mono_feat"] = feats[0]
        return out

    def forward(self, imgs, proj_matrices, depth_values, filename=None, capture=None, teacher=None):
        """``capture``: optional dict that receives per-stage cor_feats/logits.
        ``teacher``: optional dict stage-name -> hypo_depth to force (teacher
        forcing for tie-aware depth parity)."""
        dmin, dmax = depth_values[:, 0], depth_values[:, -1]
        depth_interval = (dmax - dmin) / depth_values.size(1)
        pyramids = [self.feature(img) for img in imgs]
        outputs = {}
        prev = None
        for s in range(self.num_stage):
            name = "stage%d" % (s + 1)
            feats = [p[name] for p in pyramids]
            B, C, H, W = feats[0].shape
            if teacher is not None and name in teacher:
                hypo = teacher[name]
            elif s == 0:
                hypo = (init_inverse_range if self.inverse_depth else init_range)(depth_values, self.stage_splits[s], H, W)
            elif self.inverse_depth:
                hypo = schedule_inverse_range(prev["inverse_min_depth"].detach(), prev["inverse_max_depth"].detach(),
                                              self.stage_splits[s], H, W)
            else:
                hypo = schedule_range(prev["depth"].detach(), self.stage_splits[s],
                                      self.depth_interals_ratio[s] * depth_interval, H, W)
            cap = None
            if capture is not None:
                cap = capture.setdefault(name, {})
            prev = self.run_stage(feats, proj_matrices[name], hypo, s, capture=cap)
            outputs[name] = prev
            outputs.update(prev)
        if self.mono and self.training:
            outputs = self.mono_depth_decoder(outputs, depth_values[:, 0], depth_values[:, 1])
        return outputs


# --------------------------------------------------------------------------- #
# losses (section 8f "next"; restated so training parity can be pinned)        #
# --------------------------------------------------------------------------- #
def sinkhorn(gt_depth, hypo_depth, attn_weight, mask, iters, eps=1, continuous=False):
    """Entropy-regularised OT between the ground-truth depth and attn_weight.
    Follows models/mvs4net_utils.py:1096-1142: the discrete branch (:1105-1110) transports onto the one-hot bin of
    the nearest hypothesis with the |i-j| cost; the continuous branch (:1111-1123) adds a (D+1)-th target column
    that holds all the mass and whose cost is the distance of every bin to the ground truth's fractional bin
    position (10 on masked-out pixels)."""
    B, D, H, W = attn_weight.shape
    dev = gt_depth.device
    bins = torch.arange(D, dtype=torch.float32, device=dev)
    absdiff = (bins[None, :] - bins[:, None]).abs()                                   # |i - j|
    if not continuous:
        cost = absdiff[None, None].repeat(B, H * W, 1, 1)
        gt_idx = torch.abs(hypo_depth - gt_depth[:, None]).min(1)[1].reshape(B * H * W, 1)
        gt = torch.zeros_like(hypo_depth).permute(0, 2, 3, 1).reshape(B * H * W, D)
        gt.scatter_add_(1, gt_idx, torch.ones([gt.shape[0], 1], dtype=gt.dtype, device=dev))
        gt = gt.reshape(B, H * W, D)
    else:
        gt = torch.zeros((B, H * W, D + 1), dtype=torch.float32, device=dev)
        gt[:, :, -1] = 1
        itv = 1 / hypo_depth[:, 2] - 1 / hypo_depth[:, 1]
        pos = (1 / gt_depth - 1 / hypo_depth[:, 0]) / itv                             # fractional bin of the GT
        pos[~mask] = 10
        last = (pos.unsqueeze(-1) - bins.view(1, 1, 1, D)).abs()                      # [B,H,W,D]
        cost = torch.cat([absdiff.view(1, 1, 1, D, D).expand(B, H, W, D, D), last.unsqueeze(-1)], dim=-1)
        cost = cost.reshape(B, H * W, D, D + 1)
    pred = attn_weight.permute(0, 2, 3, 1).reshape(B, H * W, D)
    log_mu = (gt + 1e-12).log()
    log_nu = (pred + 1e-12).log()
    u, v = torch.zeros_like(log_nu), torch.zeros_like(log_mu)
    for _ in range(iters):
        v = log_mu - torch.logsumexp(cost / eps + u.unsqueeze(3), dim=2)
        u = log_nu - torch.logsumexp(cost / eps + v.unsqueeze(2), dim=3)
    T = (cost / eps + u.unsqueeze(3) + v.unsqueeze(2)).exp()
    loss = (T * cost).reshape(B * H * W, -1)[mask.reshape(-1)].sum(-1).mean()
    return T, loss


def mvs4net_loss(inputs, depth_gt_ms, mask_ms, **kw):
    """Follows models/MVS4Net.py:113-155."""
    stage_lw = kw.get("stage_lw", [1, 1, 1, 1])
    l1ot_lw = kw.get("l1ot_lw", [0, 1])
    inverse = kw.get("inverse_depth", False)
    ot_iter = kw.get("ot_iter", 3)
    ot_eps = kw.get("ot_eps", 1)
    ot_continous = kw.get("ot_continous", False)
    mono = kw.get("mono", False)
    dev = mask_ms["stage1"].device
    total = torch.tensor(0.0, dtype=torch.float32, device=dev)
    l1s, ots, ranges = [], [], []
    keys = [k for k in inputs.keys() if "stage" in k]
    for si, key in enumerate(keys):
        st = inputs[key]
        hypo, attn = st["hypo_depth"], st["attn_weight"]
        mask = mask_ms[key] > 0.5
        gt = depth_gt_ms[key]
        if mono and si != 0:
            l1 = F.l1_loss(st["mono_depth"][mask], gt[mask], reduction="mean")
        else:
            l1 = torch.tensor(0.0, dtype=torch.float32, device=dev)
        if inverse:
            itv = (1 / hypo[:, 2] - 1 / hypo[:, 1]).abs()
            oor = ((1 / hypo - 1 / gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)).sum(1) == 0
        else:
            itv = (hypo[:, 2] - hypo[:, 1]).abs()
            oor = ((hypo - gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)).sum(1) == 0
        ranges.append(oor[mask].float().mean())
        ot = sinkhorn(gt, hypo, attn, mask, iters=ot_iter, eps=ot_eps, continuous=ot_continous)[1]
        l1s.append(l1)
        ots.append(ot)
        total = total + stage_lw[si] * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot)
    return total, l1s, ots, ranges


def blend_loss(inputs, depth_gt_ms, mask_ms, **kw):
    """Follows models/MVS4Net.py:158-206: MVS4net_loss plus the last stage's normalised end-point error and its
    <=3 / <=1 inlier percentages (depths scaled by 128 / (depth_max - depth_min))."""
    total, l1s, ots, ranges = mvs4net_loss(inputs, depth_gt_ms, mask_ms, **kw)
    key = [k for k in inputs.keys() if "stage" in k][-1]
    scale = 128 / (kw.get("depth_max", 100) - kw.get("depth_min", 1))[:, None, None]
    m = mask_ms[key] > 0.5
    err = ((inputs[key]["depth"] * scale)[m] - (depth_gt_ms[key] * scale)[m]).abs()
    return total, l1s, ots, ranges, err.mean(), (err <= 3).float().mean() * 100, (err <= 1).float().mean() * 100
