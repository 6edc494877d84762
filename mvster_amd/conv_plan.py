"""Host side of the MFMA convolution kernels (mvster_amd/csrc/conv_mfma.hip).

A ``ConvLayer`` owns what one fused layer needs on the device: weights packed into
MFMA fragment order, BatchNorm folded into per-channel scale/shift (eval mode), the
geometry record and the tile shape.  ``Reg2dPlan`` / ``Reg3dPlan`` / ``FpnPlan`` chain
layers exactly like the reference modules (models/mvs4net_utils.py:870-965, :419-502)
on channels-last activations.  Plans are built from the nn.Module tree of
``mvster_amd.modules`` and are rebuilt when its parameters change.
"""
import ctypes

import numpy as np
import torch

from . import _lib, ops

# geometry record shared with mvster_conv_mfma (int32, host memory)
GEOM = ("B", "Di", "Hi", "Wi", "Do", "Ho", "Wo", "DoF", "HoF", "WoF", "sd", "sh", "sw", "cout", "ntile_total", "relu",
        "skip_mode", "osd", "osh", "osw", "nclass")
GEOM_CLASS = ("kd", "kh", "kw", "pd", "ph", "pw", "od", "oh", "ow", "nsteps")

SKIP_NONE, SKIP_ADD, SKIP_UPSAMPLE_ADD = 0, 1, 2


def _pack_gemm(wk):
    """[K, N] -> fragment order [K/16, N/16, 64 lanes, 4] (K and N zero-padded to multiples of 16).
    Lane (q = lane >> 4, n = lane & 15) of K-step s and N-tile t holds wk[s*16 + q*4 + j, t*16 + n], j=0..3."""
    K, N = wk.shape
    Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
    full = wk.new_zeros(Kp, Np)
    full[:K, :N] = wk
    return full.view(Kp // 16, 4, 4, Np // 16, 16).permute(0, 3, 1, 4, 2).contiguous().view(-1), Kp // 16


def fold_bn(bn, cout, device):
    """eval BatchNorm -> (scale, shift); None -> identity."""
    if bn is None:
        return torch.ones(cout, device=device), torch.zeros(cout, device=device)
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    shift = bn.bias.detach() - bn.running_mean.detach() * scale
    return scale.float(), shift.float()


# Measured per-layer choices for gfx950 (scripts/conv_microbench.py --emit): signature -> [variant, mt, nt].
# A missing entry falls back to the heuristics below, so the table only ever changes speed.
_TUNING = None


def _tuning():
    global _TUNING
    if _TUNING is None:
        import json
        import os
        path = os.environ.get("MVSTER_TUNING") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning_gfx950.json")
        try:
            with open(path) as f:
                _TUNING = json.load(f)
        except (OSError, ValueError):
            _TUNING = {}
    return _TUNING


def layer_signature(layer, B, Di, Hi, Wi, skip_mode):
    return "%s%d-%d_k%dx%dx%d_s%dx%dx%d_%dx%dx%dx%d_sk%d" % (
        "T" if layer.transposed else "C", layer.cin, layer.cout, *layer.kernel, *layer.stride, B, Di, Hi, Wi, skip_mode)


# The measured table is keyed on exact shapes; any other resolution used to fall straight to the heuristics.  Its choices
# are a function of the layer FAMILY (channels, kernel, stride, skip mode) and of the map's size and depth far more than of
# the exact shape, so an unknown shape takes the choice of the family's entry nearest in
#     |log2 voxels ratio| + |log2 depth ratio| + 0.25 |log2 batch ratio|
# (the depth split matters to the 3-D kernels and to how many slices share a tile row: 4 x 48 x 64 and 8 x 32 x 48 voxels
# want different ring-kernel modes) within FAMILY_REACH before the heuristics; exact entries override.
FAMILY_REACH = 2.0
_FAMILIES = None
_SIG = None


def _families():
    global _FAMILIES, _SIG
    if _FAMILIES is None:
        import math
        import re
        _SIG = re.compile(r"^([CT]\d+-\d+_k\dx\dx\d_s\dx\dx\d)_(\d+)x(\d+)x(\d+)x(\d+)_sk([012])$")
        fam = {}
        for sig, val in _tuning().items():
            m = _SIG.match(sig)
            if not m:
                continue
            B, D, H, W = (int(m.group(i)) for i in (2, 3, 4, 5))
            fam.setdefault((m.group(1), int(m.group(6))), []).append((math.log2(B * D * H * W), math.log2(D), math.log2(B), list(val)))
        for lst in fam.values():
            lst.sort(key=lambda e: e[:3])
        _FAMILIES = fam
    return _FAMILIES


def tuned_choice(layer, B, Di, Hi, Wi, skip_mode):
    """([variant word, mt, nt], "exact" | "family") from the measured table, or (None, None)."""
    import math
    sig = layer_signature(layer, B, Di, Hi, Wi, skip_mode)
    hit = _tuning().get(sig)
    if hit:
        return hit, "exact"
    if skip_mode == SKIP_ADD:
        # a same-shape skip only adds one read to the epilogue: where the table knows the layer without it (the training
        # step's input-gradient layers that add a second consumer's gradient, train_ops.conv_cl(tap=True)), that choice
        # beats the heuristics (a kernel form without the skip path answers "unsupported" and the caller falls back)
        hit = _tuning().get(sig[:-1] + "0")
        if hit:
            return hit, "exact (without skip)"
    lst = _families().get((sig[:sig.index("_", sig.index("_s") + 1)], skip_mode))
    if lst:
        lv, ld, lb = math.log2(max(1, B * Di * Hi * Wi)), math.log2(max(1, Di)), math.log2(max(1, B))
        d, val = min(((abs(e[0] - lv) + abs(e[1] - ld) + 0.25 * abs(e[2] - lb), e[3]) for e in lst), key=lambda t: t[0])
        if d <= FAMILY_REACH:
            return list(val), "family"
    return None, None


FUSE_TAIL = True                # finest FPN level: lateral step + gather-sum in one launch (tests set it to False for the two launches)
FUSE_SELECT = True              # reg2d conv11 + prob + selection in one launch (tests set it to False for the two-launch form)
LDS_BUDGET = 12 * 256 * 16      # bytes of staged patch the LDS variant accepts (conv_mfma.hip kMaxStage)
FORCE_VARIANT = None            # None = choose per layer; 0 / 1 pin the kernel variant (experiments, tests)
TPERS16_MIN_VOXELS = 40960      # 16 -> 8 transposed layers take the persistent MFMA kernel from this many input voxels
FUSE_CONV0 = True               # FPN conv0[0] -> conv0[1] in one launch (FpnPlan._conv0)
NARROW_PAIR_WPC = 1             # ... workgroups per CU of that launch (0 = the kernel's default, 2): alone 46-48 against 45 us, but with two depth maps in flight 1 169 against 1 156 depth-maps/s -- half of the LDS stays free for the other forward's kernels
NARROW_PAIR_MIN_PIXELS = 512 * 14 * 64   # ... from two 14 x 64 tiles per CU (below: the two layers' own launches)
NARROW_MIN_VOXELS = 64 * 256    # untuned narrow layers take the MFMA kernel from this many output voxels (one 8 x 32 tile per CU)


def _lds_plan(B, Do, Ho, Wo, kernel, stride, ntile_total):
    """(mt, nt) for the LDS-staged kernel, or None if the layer does not fit it."""
    kd, kh, kw = kernel
    if kw not in (3, 5) or Wo < 24:
        return None
    for mt in (4, 2):
        ty = 2 * mt
        patch = kd * ((ty - 1) * stride[1] + kh) * (31 * stride[2] + kw) * 64
        if patch > LDS_BUDGET or (mt == 4 and Ho < 6):
            continue
        blocks = -(-Wo // 32) * -(-Ho // ty) * Do * B
        nt = max(n for n in (4, 2, 1) if n <= ntile_total and ntile_total % n == 0)
        while nt > 1 and blocks * (ntile_total // nt) < 512:
            nt //= 2
        return mt, nt
    return None


# (mode, cin / 16, kd) instances of conv_wino_ring_kernel (dispatch_wino, conv_wino.hip): mode 0 = one N tile, waves 4-7 load;
# 1 = the two halves of the workgroup compute one N tile each; 2 = the compute waves hold both N tiles, waves 4-7 load
WINO_RING_INSTANCES = {(0, 1, 3), (0, 2, 3), (1, 2, 3), (2, 2, 3), (0, 4, 3), (1, 4, 3), (2, 4, 3),
                       (0, 4, 1), (1, 4, 1), (2, 4, 1), (0, 2, 1), (1, 2, 1), (2, 2, 1), (0, 1, 1), (2, 1, 1)}
# (nt, cin / 16) instances of conv_wino_kernel (1 x 3 x 3 only)
WINO_INSTANCES = {(1, 1), (2, 1), (2, 2), (1, 2)}


def _wino_plan(layer, B, Do, Ho, Wo):
    """(variant, mt, nt) of the Winograd kernel for an eligible layer, or None when the map is too small to fill the chip
    with 8 x 32 tiles (the split-K / direct kernels are faster there: scripts/conv_wino_check.py)."""
    tiles = B * Do * -(-Ho // 8) * -(-Wo // 32)
    kd = layer.kernel[0]
    if kd == 1 and layer.cin in (16, 32):
        nt = 2 if layer.ntile_total % 2 == 0 else 1
        if tiles * (layer.ntile_total // nt) >= 192:
            return 8 | (1 << 8), 2, nt
        return None
    nt = 2 if (layer.ntile_total % 2 == 0 and tiles * (layer.ntile_total // 2) >= 224) else 1
    if (1, layer.cin // 16, kd) not in WINO_RING_INSTANCES:
        nt = 1                          # (16 input channels: the ring kernel has its one-N-tile form only)
    if tiles * (layer.ntile_total // nt) >= 160 and (0 if nt == 1 else 1, layer.cin // 16, kd) in WINO_RING_INSTANCES:
        return 9, 2, nt
    return None


def _tiles(M, ntile_total, nclass):
    """(MT, NT): biggest register tile that still fills the chip (>= 2048 waves), else the most waves."""
    best = None
    for nt in (5, 4, 3, 2, 1):
        if nt > ntile_total or ntile_total % nt:
            continue
        for mt in (4, 2, 1):
            waves = -(-M // (16 * mt)) * (ntile_total // nt) * nclass
            if waves >= 2048:
                return mt, nt
            if best is None or waves > best[0]:
                best = (waves, mt, nt)
    return best[1], best[2]


class ConvLayer:
    """One fused conv (+scale/shift, +ReLU, +skip) layer on channels-last tensors."""

    def __init__(self, weight, transposed, stride, padding, bn=None, bias=None, relu=False, cin_pad=None, prob=None):
        w = weight.detach().float()
        dev = w.device
        if w.dim() == 4:  # 2-D conv -> depth 1
            w = w.unsqueeze(2)
            stride = (1,) + tuple(stride)
            padding = (0,) + tuple(padding)
        self.transposed = transposed
        self.stride = tuple(stride)
        self.padding = tuple(padding)
        if transposed:
            cin, cout = w.shape[0], w.shape[1]
        else:
            cout, cin = w.shape[0], w.shape[1]
        self.cin = cin_pad or cin
        if self.cin not in (4, 8, 16, 32, 64, 80):        # (80: the FPN gather's 72 gradient channels, padded)
            raise RuntimeError("conv_mfma: unsupported input channel count %d" % self.cin)
        self.cout = cout
        self.relu = relu
        kd, kh, kw = w.shape[2:]
        self.kernel = (kd, kh, kw)
        self._cin_raw = cin
        self.classes, self.woff = self._class_table()
        self.wpk = None
        self.w_small = self.w_small4 = self.w_deconv = self.wpk_wino = self.wpk_b3 = self._b3_src = None
        self._pack(w)
        self.ntile_total = (cout + 15) // 16
        npad = self.ntile_total * 16
        scale, shift = fold_bn(bn, cout, dev)
        if bias is not None:
            shift = shift + bias.detach().float() * scale
        self.scale = torch.zeros(npad, device=dev)
        self.shift = torch.zeros(npad, device=dev)
        self.scale[:cout] = scale
        self.shift[:cout] = shift
        self.zeros = torch.zeros(64, device=dev)
        self._geom_cache = {}
        # optional fused 1x1x1 head: (weight [8], bias [1]) -> the layer outputs logits [B,D,H,W]
        self.prob = None
        if prob is not None:
            if cout != 8:
                raise RuntimeError("conv_mfma: the fused prob head needs 8 channels")
            self.prob = (prob[0].detach().float().reshape(-1).contiguous().to(dev),
                         prob[1].detach().float().reshape(-1).contiguous().to(dev))
    @staticmethod
    def _axis_classes(k, s, p):
        """Output-phase classes of one axis of a transposed conv: [(phase, [kernel taps in input-offset
        order], padding)].  Output o = s*i + phase gets input i + d through tap k with o = s*i' - p + k,
        i.e. d = (phase + p - k) / s; the class is an ordinary stride-1 conv over the input lattice with
        those taps and padding -min(d).  (3, 2, 1) and (5, 2, 2) double the size (output_padding 1): the
        reference's up-convolutions and the input gradients of its stride-2 3x3 / 5x5 convolutions."""
        if s == 1:
            if k != 1 or p != 0:
                raise RuntimeError("conv_mfma: transposed stride-1 axis must have kernel 1")
            return [(0, [0], 0)]
        if (k, s, p) not in ((3, 2, 1), (5, 2, 2)):
            raise RuntimeError("conv_mfma: transposed conv must be k=3, s=2, p=1 or k=5, s=2, p=2")
        out = []
        for phase in range(s):
            taps = sorted(((phase + p - kk) // s, kk) for kk in range(k) if (phase + p - kk) % s == 0)
            out.append((phase, [kk for _, kk in taps], -taps[0][0]))
        return out

    def _class_table(self):
        """Weight-independent part of the plan: the GEOM_CLASS records and the packed-weight offsets."""
        kd, kh, kw = self.kernel
        if not self.transposed:
            nsteps = (kd * kh * kw * self.cin + 15) // 16
            return [dict(kd=kd, kh=kh, kw=kw, pd=self.padding[0], ph=self.padding[1], pw=self.padding[2], od=0, oh=0,
                         ow=0, nsteps=nsteps, taps=None)], np.zeros(1, dtype=np.int64)
        classes, woff, off = [], [], 0
        npad = (self.cout + 15) // 16 * 16
        for pz, kzs, qz in self._axis_classes(kd, self.stride[0], self.padding[0]):
            for py, kys, qy in self._axis_classes(kh, self.stride[1], self.padding[1]):
                for px, kxs, qx in self._axis_classes(kw, self.stride[2], self.padding[2]):
                    nsteps = (len(kzs) * len(kys) * len(kxs) * self.cin + 15) // 16
                    classes.append(dict(kd=len(kzs), kh=len(kys), kw=len(kxs), pd=qz, ph=qy, pw=qx, od=pz, oh=py,
                                        ow=px, nsteps=nsteps, taps=(kzs, kys, kxs)))
                    woff.append(off)
                    off += nsteps * 16 * npad
        return classes, np.asarray(woff, dtype=np.int64)

    def _pack(self, w):
        """(Re)build every device-side weight form from ``w`` (torch ops only, so it also runs on the device
        once per optimizer step in training)."""
        dev = w.device
        kd, kh, kw = self.kernel
        cin, cout = self._cin_raw, self.cout
        packed = []
        if not self.transposed:
            wk = w.permute(2, 3, 4, 1, 0)  # [kd,kh,kw,cin,cout]
            if self.cin != cin:
                wk = torch.nn.functional.pad(wk, (0, 0, 0, self.cin - cin))
            packed.append(_pack_gemm(wk.reshape(kd * kh * kw * self.cin, cout))[0])
        else:
            if self.cin != cin:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, 0, 0, self.cin - cin))
            wf = w.reshape(w.shape[0], w.shape[1], kd * kh * kw)
            for c in self.classes:
                # the class's taps as one device index over the flattened (kz, ky, kx) axis; built once per layer and
                # device: indexing with Python lists would upload an index tensor (a synchronising copy) per call
                idx = c.get("tap_index")
                if idx is None or idx.device != dev:
                    kzs, kys, kxs = c["taps"]
                    idx = c["tap_index"] = torch.tensor([(z * kh + y) * kw + x for z in kzs for y in kys for x in kxs],
                                                        dtype=torch.long, device=dev)
                sub = wf.index_select(2, idx)                       # [cin,cout,|kz|*|ky|*|kx|]
                packed.append(_pack_gemm(sub.permute(2, 0, 1).reshape(-1, cout))[0])
        self.wpk = torch.cat(packed).contiguous()
        # HBM-bound finest up-sampling layers: VALU kernel (deconv_small), weights as [3,3,cin,cout]
        self.w_deconv = None
        if (self.transposed and self.kernel == (1, 3, 3) and self.stride == (1, 2, 2) and self.padding == (0, 1, 1)
                and (self.cin, cout) in ((16, 8),)):      # (32, 16) measured slower than the MFMA classes
            self.w_deconv = w[:, :, 0].permute(2, 3, 0, 1).contiguous().to(dev)     # [3,3,cin,cout]
        # narrow full-resolution layers: VALU kernel (conv_small.hip), weights as [3,3,cin,8]
        self.w_small = None
        if (not self.transposed and self.kernel == (1, 3, 3) and self.stride == (1, 1, 1) and self.padding == (0, 1, 1)
                and cout == 8 and self.cin in (4, 8)):
            ws = w[:, :, 0].permute(2, 3, 1, 0)                       # [3,3,cin,8]
            if self.cin != cin:
                ws = torch.nn.functional.pad(ws, (0, 0, 0, self.cin - cin))
            self.w_small = ws.contiguous().to(dev)
        # 8 -> 4 (the input gradient of reg2d's first layer): the narrow MFMA kernel's four-channel form, weights as [3,3,8,8]
        # with zero output columns 4..7
        self.w_small4 = None
        if (not self.transposed and self.kernel == (1, 3, 3) and self.stride == (1, 1, 1) and self.padding == (0, 1, 1)
                and cout == 4 and self.cin == 8):
            ws = w[:, :, 0].permute(2, 3, 1, 0)                       # [3,3,cin,4]
            ws = torch.nn.functional.pad(ws, (0, 4, 0, self.cin - cin))
            self.w_small4 = ws.contiguous().to(dev)
        self._pack_wino(w)
        self._pack_b3(w)

    def b3_eligible(self):
        """(1|3)x3x3 stride-1 layers of 16 / 32 / 64 channels that the bf16-split kernel (variant 11, conv_b3.hip) covers."""
        return (not self.transposed and self.kernel in ((1, 3, 3), (3, 3, 3)) and self.stride == (1, 1, 1)
                and self.padding == (self.kernel[0] // 2, 1, 1) and self.cin in (16, 32, 64) and self.cout in (16, 32, 64))

    def _pack_b3(self, w):
        """The bf16-split weight fragments of an eligible layer are built on first use (variant 11 is a probe: the plan
        never selects it); here only the source is remembered (no copy: the parameter's own storage when it is fp32)."""
        self.wpk_b3 = None
        self._b3_src = w if (self.b3_eligible() and w.is_cuda) else None

    def wino_eligible(self):
        """(1|3)x3x3 stride-1 layers the Winograd kernels (variants 8 / 9, conv_wino.hip) cover."""
        return (not self.transposed and self.kernel in ((1, 3, 3), (3, 3, 3)) and self.stride == (1, 1, 1)
                and self.padding == (self.kernel[0] // 2, 1, 1) and self.cin in (16, 32, 64) and self.cout % 16 == 0)

    def _pack_wino(self, w, swap=False, flip=False):
        """Transformed weights G g G^T of an eligible layer in the packed fragment order (one launch, reads ``w`` in place)."""
        if not self.wino_eligible() or not w.is_cuda:
            self.wpk_wino = None
            return
        if self.wpk_wino is None:
            self.wpk_wino = torch.empty(self.kernel[0] * 16 * self.cin * ((self.cout + 15) // 16) * 16, device=w.device,
                                        dtype=torch.float32)
        st = w.stride()
        s_n, s_c = (st[1], st[0]) if swap else (st[0], st[1])
        rc = _lib.load().mvster_pack_wino_weights(w.data_ptr(), self.wpk_wino.data_ptr(), self.cout, self._cin_raw, self.cin,
                                                  self.kernel[0], s_n, s_c, st[2], st[3], st[4], int(flip), ops._stream())
        _lib.check(rc, "pack_wino_weights")

    def repack_on_device(self, weight, swap=False, flip=False):
        """Refresh ``wpk`` of an ordinary (non-transposed) layer with one kernel, reading the parameter tensor in
        place.  ``swap``: the tensor's dim 1 is this layer's output channel (a ConvTranspose weight, or the
        input-gradient form of a conv); ``flip``: mirror the taps (input-gradient form of a stride-1 conv)."""
        w = weight.detach()
        if w.dim() == 4:
            w = w.unsqueeze(2)
        if w.dtype != torch.float32 or not w.is_cuda:
            raise RuntimeError("repack_on_device: fp32 CUDA parameter expected")
        if self.transposed:
            return self._repack_classes_on_device(w)
        st = w.stride()
        s_n, s_c = (st[1], st[0]) if swap else (st[0], st[1])
        kd, kh, kw = self.kernel
        rc = _lib.load().mvster_pack_conv_weights(w.data_ptr(), self.wpk.data_ptr(), self.cout, self._cin_raw, self.cin, kd,
                                                  kh, kw, s_n, s_c, st[2], st[3], st[4], int(flip),
                                                  ops._stream())
        _lib.check(rc, "pack_conv_weights")
        self.wpk_b3 = self._b3_src = None          # (probe variant 11: not for layers whose weights are refreshed in place)
        if self.wpk_wino is not None:
            self._pack_wino(w, swap, flip)
        if self.w_small is not None:
            wv = w.transpose(0, 1) if swap else w
            if flip:
                wv = wv.flip(2, 3, 4)
            ws = wv[:, :, 0].permute(2, 3, 1, 0)                      # [3,3,cin,8]
            if self.cin != self._cin_raw:
                ws = torch.nn.functional.pad(ws, (0, 0, 0, self.cin - self._cin_raw))
            self.w_small.copy_(ws)
        if getattr(self, "w_small4", None) is not None:
            wv = w.transpose(0, 1) if swap else w
            if flip:
                wv = wv.flip(2, 3, 4)
            self.w_small4[:, :, :self._cin_raw, :4].copy_(wv[:, :, 0].permute(2, 3, 1, 0))      # [3,3,cin,4]; the rest stays zero

    def _repack_classes_on_device(self, w):
        """Transposed layer: every output-parity class in one launch (tap lists and block offsets from the class table)."""
        kd, kh, kw = self.kernel
        if not w.is_contiguous() or tuple(w.shape[:2]) != (self._cin_raw, self.cout):
            raise RuntimeError("repack_on_device: contiguous [cin, cout, kd, kh, kw] weight expected")
        tab = getattr(self, "_class_pack", None)
        if tab is None:
            ntaps = np.zeros(len(self.classes), dtype=np.int32)
            taps = np.zeros((len(self.classes), 27), dtype=np.int32)
            for i, c in enumerate(self.classes):
                kzs, kys, kxs = c["taps"]
                flat = [(z * kh + y) * kw + x for z in kzs for y in kys for x in kxs]
                ntaps[i] = len(flat)
                taps[i, :len(flat)] = flat
            tab = self._class_pack = (ntaps, np.ascontiguousarray(self.woff, dtype=np.int64), taps)
        ntaps, woff, taps = tab
        rc = _lib.load().mvster_pack_conv_weights_classes(
            w.data_ptr(), self.wpk.data_ptr(), self.cout, self._cin_raw, self.cin, kd * kh * kw, len(self.classes),
            ntaps.ctypes.data_as(ctypes.c_void_p), woff.ctypes.data_as(ctypes.c_void_p), taps.ctypes.data_as(ctypes.c_void_p),
            ops._stream())
        _lib.check(rc, "pack_conv_weights_classes")
        if self.w_deconv is not None:
            self.w_deconv.copy_(w[:, :, 0].permute(2, 3, 0, 1))

    def repack(self, weight, bias=None):
        """Refresh the packed weights (and the bias folded into ``shift``) after a parameter update; geometry,
        tile choices and launch records are kept."""
        w = weight.detach().float()
        if w.dim() == 4:
            w = w.unsqueeze(2)
        self._pack(w)
        if bias is not None:
            self.shift[:self.cout] = bias.detach().float() * self.scale[:self.cout]

    def out_shape(self, B, Di, Hi, Wi):
        kd, kh, kw = self.kernel
        s, p = self.stride, self.padding
        if self.transposed:
            return (B, Di * s[0], Hi * s[1], Wi * s[2])
        return (B, (Di + 2 * p[0] - kd) // s[0] + 1, (Hi + 2 * p[1] - kh) // s[1] + 1, (Wi + 2 * p[2] - kw) // s[2] + 1)

    def _geom(self, B, Di, Hi, Wi, skip_mode):
        key = (B, Di, Hi, Wi, skip_mode)
        g = self._geom_cache.get(key)
        if g is None:
            _, DoF, HoF, WoF = self.out_shape(B, Di, Hi, Wi)
            s = self.stride
            if self.transposed:
                Do, Ho, Wo = Di, Hi, Wi
                sd = sh = sw = 1
                osd, osh, osw = s
            else:
                Do, Ho, Wo = DoF, HoF, WoF
                sd, sh, sw = s
                osd = osh = osw = 1
            vals = dict(B=B, Di=Di, Hi=Hi, Wi=Wi, Do=Do, Ho=Ho, Wo=Wo, DoF=DoF, HoF=HoF, WoF=WoF, sd=sd, sh=sh, sw=sw,
                        cout=self.cout, ntile_total=self.ntile_total, relu=int(self.relu), skip_mode=skip_mode,
                        osd=osd, osh=osh, osw=osw, nclass=len(self.classes))
            arr = [vals[k] for k in GEOM]
            for c in self.classes:
                arr += [c[k] for k in GEOM_CLASS]
            mt, nt = _tiles(B * Do * Ho * Wo, self.ntile_total, len(self.classes))
            variant = 0
            lds = None
            if not self.transposed and self.cin % 16 == 0 and skip_mode in (SKIP_NONE, SKIP_ADD):
                lds = _lds_plan(B, Do, Ho, Wo, self.kernel, self.stride, self.ntile_total)
            if lds is not None and FORCE_VARIANT != 0:
                variant, (mt, nt) = 1, lds
            if variant == 0 and self.cin >= 16 and FORCE_VARIANT is None:
                # small deep layers: let the 4 waves of a workgroup split K instead of M (variant 2)
                tiles16 = -(-(B * Do * Ho * Wo) // 16)
                waves = -(-tiles16 // mt) * (self.ntile_total // nt) * len(self.classes)
                if waves < 1024 and min(c["nsteps"] for c in self.classes) >= 8:
                    variant, mt = 2, 1
                    nt = 1
                    for cand in (4, 2):
                        if self.ntile_total % cand == 0 and tiles16 * (self.ntile_total // cand) * len(self.classes) >= 512:
                            nt = cand
                            break
            if (FORCE_VARIANT is None and self.wpk_wino is not None and self.prob is None
                    and skip_mode in (SKIP_NONE, SKIP_ADD)):
                # shapes the measured table does not know: the Winograd kernels win wherever there is a work unit
                # (8 x 32 output pixels x one pair of N tiles) for most CUs -- the pattern of the table's 61 entries
                wv = _wino_plan(self, B, Do, Ho, Wo)
                if wv is not None:
                    variant, mt, nt = wv
            tuned = tuned_choice(self, B, Di, Hi, Wi, skip_mode)[0] if FORCE_VARIANT is None else None
            if tuned and (tuned[0] & 0xff) in (8, 9) and (self.wpk_wino is None or self.prob is not None):
                # a Winograd entry measured on a neighbouring layer (padding and the fused prob head are not part of the
                # signature): this layer has no transformed weights -- keep the heuristic choice instead of raising later
                tuned = None
            if tuned:
                variant, mt, nt = tuned
            if self.w_small is not None and skip_mode in (SKIP_NONE, SKIP_ADD) and FORCE_VARIANT in (None, 3, 10):
                # narrow layers: shift-packed MFMA tiles on the persistent LDS-DMA ring (variant 10, conv_narrow.hip; mt = tile
                # rows / 4, nt = workgroups per CU, 0 = the kernel's defaults) or the VALU kernel (3) for maps too small to
                # give every CU a tile
                if FORCE_VARIANT is not None:
                    variant, mt, nt = FORCE_VARIANT, 0, 0
                elif tuned and tuned[0] in (3, 10):
                    variant, mt, nt = tuned
                else:
                    variant, mt, nt = (10, 0, 0) if B * Do * Ho * Wo >= NARROW_MIN_VOXELS else (3, 0, 0)
                if variant == 10 and not hasattr(_lib.load(), "mvster_conv_narrow"):
                    variant = 3                     # (an older library loaded through MVSTER_LIB for an A/B run)
            if (getattr(self, "w_small4", None) is not None and skip_mode == SKIP_NONE and FORCE_VARIANT is None
                    and self.prob is None and B * Do * Ho * Wo >= NARROW_MIN_VOXELS and hasattr(_lib.load(), "mvster_conv_narrow4")):
                variant, mt, nt = 10, 0, 0          # (129 -> 60 us at [2, 4, 512, 640]: it ran on the direct kernel)
            if self.w_deconv is not None and skip_mode in (SKIP_NONE, SKIP_ADD) and FORCE_VARIANT in (None, 4):
                variant = 4
                if (FORCE_VARIANT is None and self.prob is None and self.cin == 16 and self.cout == 8
                        and B * Di * Hi * Wi >= TPERS16_MIN_VOXELS):
                    # 16 -> 8 transposed 3x3 stride 2 without the fused head (the training step's conv11 and the input
                    # gradient of conv1): the persistent MFMA kernel of the wider transposed layers, 41 against 55 us at
                    # 2 x 4 x 256 x 320, 15 against 22 at 2 x 4 x 128 x 160 (the VALU kernel is instruction-bound there)
                    variant, mt, nt = 5, 2, 1
            if (FORCE_VARIANT is None and self.transposed and self.kernel == (1, 5, 5) and self.stride == (1, 2, 2)
                    and self.padding == (0, 2, 2) and self.cin in (16, 32) and self.prob is None
                    and skip_mode in (SKIP_NONE, SKIP_ADD) and B * Di * Hi * Wi >= TPERS16_MIN_VOXELS):
                # the input gradients of the FPN's 5x5 stride-2 convolutions (training): the persistent kernel's 1x5x5 form,
                # bit-identical to the direct kernel the table's entries name; 16 -> 8 at [10, 256, 320] 200 -> 109 us,
                # 32 -> 16 at [10, 128, 160] 113 -> 66 us (scripts/conv_tpers5_check.py)
                variant, mt, nt = 5, 2, 1
            g = (np.asarray(arr, dtype=np.int32), mt, nt, (B, DoF, HoF, WoF), variant)
            self._geom_cache[key] = g
        return g

    def __call__(self, x, skip=None, skip_mode=SKIP_NONE, tiles=None):
        """x [B,Di,Hi,Wi,cin] channels-last -> [B,Do,Ho,Wo,cout].  ``tiles`` = (mt, nt[, variant]) overrides
        the per-layer choice (tests, tuning)."""
        B, Di, Hi, Wi, C = x.shape
        if C != self.cin or not x.is_contiguous() or x.dtype != torch.float32 or not x.is_cuda:
            raise RuntimeError("conv_mfma: bad input (shape %s, cin %d)" % (tuple(x.shape), self.cin))
        if skip is None:
            skip_mode = SKIP_NONE
        geom, mt, nt, oshape, variant = self._geom(B, Di, Hi, Wi, skip_mode)
        if tiles is not None:
            mt, nt = tiles[0], tiles[1]
            variant = tiles[2] if len(tiles) > 2 else 0
        if variant == 4:
            if self.w_deconv is None or skip_mode == SKIP_UPSAMPLE_ADD:
                raise RuntimeError("deconv_small: layer not eligible")
            out = torch.empty(oshape + ((self.cout,) if self.prob is None else ()), device=x.device, dtype=torch.float32)
            if skip is not None and tuple(skip.shape) != tuple(oshape + (self.cout,)):
                raise RuntimeError("deconv_small: skip shape %s" % (tuple(skip.shape),))
            rc = _lib.load().mvster_deconv_small(
                x.data_ptr(), self.w_deconv.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                None if skip is None else skip.data_ptr(),
                None if self.prob is None else self.prob[0].data_ptr(), None if self.prob is None else self.prob[1].data_ptr(),
                out.data_ptr(), B * Di, Hi, Wi, self.cin, self.cout, int(self.relu), ops._stream())
            _lib.check(rc, "deconv_small")
            return out
        if self.prob is not None and variant in (1, 3, 10):
            variant = 0
            mt, nt = _tiles(B * geom[4] * geom[5] * geom[6], self.ntile_total, len(self.classes))
        out = torch.empty(oshape + ((self.cout,) if self.prob is None else ()), device=x.device, dtype=torch.float32)
        if skip is not None:
            if not skip.is_contiguous():
                raise RuntimeError("conv_mfma: skip must be contiguous")
            want = (oshape + (self.cout,)) if skip_mode == SKIP_ADD else (B, 1, oshape[2] // 2, oshape[3] // 2, self.cout)
            if tuple(skip.shape) != tuple(want):
                raise RuntimeError("conv_mfma: skip shape %s, expected %s" % (tuple(skip.shape), tuple(want)))
        if variant == 10 and self.w_small is None and getattr(self, "w_small4", None) is not None:
            if skip is not None:
                raise RuntimeError("conv_narrow4: the four-channel form has no skip path")
            rc = _lib.load().mvster_conv_narrow4(
                x.data_ptr(), self.w_small4.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(), out.data_ptr(), B * Di, Hi, Wi,
                int(self.relu), mt if mt in (2, 4) else 0, nt & 31, ops._stream())
            _lib.check(rc, "conv_narrow4")
            return out
        if variant == 10:
            if self.w_small is None or skip_mode == SKIP_UPSAMPLE_ADD:
                raise RuntimeError("conv_narrow: layer not eligible")
            rc = _lib.load().mvster_conv_narrow(
                x.data_ptr(), self.w_small.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                None if skip is None else skip.data_ptr(), out.data_ptr(), B * Di, Hi, Wi, self.cin, int(self.relu),
                mt if mt in (2, 4) else 0, nt & 31, ops._stream())
            _lib.check(rc, "conv_narrow")
            return out
        if variant == 3:
            if self.w_small is None or skip_mode == SKIP_UPSAMPLE_ADD:
                raise RuntimeError("conv_small: layer not eligible")
            rc = _lib.load().mvster_conv_small(
                x.data_ptr(), self.w_small.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                None if skip is None else skip.data_ptr(), out.data_ptr(), B * Di, Hi, Wi, self.cin, int(self.relu),
                ops._stream())
            _lib.check(rc, "conv_small")
            return out
        wpk = self.wpk
        if (variant & 0xff) in (8, 9):
            if self.wpk_wino is None:
                raise RuntimeError("conv_wino: layer not eligible")
            wpk = self.wpk_wino
        if (variant & 0xff) == 11:
            if self._b3_src is None or skip_mode == SKIP_UPSAMPLE_ADD or self.prob is not None:
                raise RuntimeError("conv_b3: layer not eligible")
            stamp = (self._b3_src._version, self._b3_src.data_ptr())
            if self.wpk_b3 is None or getattr(self, "_b3_stamp", None) != stamp:     # (in-place weight updates re-split)
                self.wpk_b3 = pack_b3(self._b3_src, self.cin, self.cout, self.kernel[0])
                self._b3_stamp = stamp
            wpk, nt = self.wpk_b3, 1
        rc = _lib.load().mvster_conv_mfma(
            x.data_ptr(), wpk.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
            None if skip is None else skip.data_ptr(), self.zeros.data_ptr(),
            None if self.prob is None else self.prob[0].data_ptr(), None if self.prob is None else self.prob[1].data_ptr(),
            out.data_ptr(), geom.ctypes.data_as(ctypes.c_void_p), int(geom.size), self.woff.ctypes.data_as(ctypes.c_void_p), self.cin,
            mt, nt, variant, ops._stream())
        if rc == -3 and tiles is None and variant != 0:
            # a heuristic (or stale table) choice the library has no instance for: the direct kernel takes every layer
            mt, nt = _tiles(B * geom[4] * geom[5] * geom[6], self.ntile_total, len(self.classes))
            rc = _lib.load().mvster_conv_mfma(
                x.data_ptr(), self.wpk.data_ptr(), self.scale.data_ptr(), self.shift.data_ptr(),
                None if skip is None else skip.data_ptr(), self.zeros.data_ptr(),
                None if self.prob is None else self.prob[0].data_ptr(), None if self.prob is None else self.prob[1].data_ptr(),
                out.data_ptr(), geom.ctypes.data_as(ctypes.c_void_p), int(geom.size), self.woff.ctypes.data_as(ctypes.c_void_p),
                self.cin, mt, nt, 0, ops._stream())
        _lib.check(rc, "conv_mfma")
        return out

    def flops(self, B, Di, Hi, Wi):
        """Algorithmic FLOPs (2*MACs) of one call, as torch.utils.flop_counter counts them."""
        _, Do, Ho, Wo = self.out_shape(B, Di, Hi, Wi)
        kd, kh, kw = self.kernel
        if self.transposed:
            return 2 * B * Di * Hi * Wi * kd * kh * kw * self.cin * self.cout
        return 2 * B * Do * Ho * Wo * kd * kh * kw * self.cin * self.cout


def pack_b3(w, cin, cout, kd):
    """w [cout, cin_raw, kd, 3, 3] -> the weights as THREE bf16 planes (w = w1 + w2 + w3 exactly: round-to-nearest bf16 of the
    value, of the residual, of the residual's residual) in the fragment order of conv_b3.hip: per stage (kd, 16 input
    channels) and N split a block [tap pair 5][N tile NTW][plane 3][lane 64][8 bf16], zero-padded to 1024 NTW 16-byte
    units; lane (n = lane & 15, g = lane >> 4) holds channels 8 (g & 1) .. + 7 of tap 2 tp + (g >> 1) (tap 9 = 0) of output
    channel 16 (ns NTW + j) + n.  -> bf16 tensor [kd, cin / 16, nsplit, 1024 NTW * 8].  Torch ops, once per plan build."""
    if w.shape[1] != cin:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cin - w.shape[1]))
    w = w.reshape(cout, cin, kd, 9).float()
    w1 = w.to(torch.bfloat16)
    r = w - w1.float()
    w2 = r.to(torch.bfloat16)
    w3 = (r - w2.float()).to(torch.bfloat16)
    planes = torch.nn.functional.pad(torch.stack([w1, w2, w3]), (0, 1))        # [3, cout, cin, kd, 10]: tap 9 = 0
    ntw = 1 if cout == 16 else 2
    nsplit = cout // (16 * ntw)
    # [pl, ns, j, n, ch, c8, e, kd, tp, th] -> [kd, ch, ns, tp, j, pl, th, c8, n, e]
    t = planes.reshape(3, nsplit, ntw, 16, cin // 16, 2, 8, kd, 5, 2).permute(7, 4, 1, 8, 2, 0, 9, 5, 3, 6)
    t = t.reshape(kd, cin // 16, nsplit, 5 * ntw * 3 * 64 * 8)
    return torch.nn.functional.pad(t, (0, 1024 * ntw * 8 - t.shape[-1])).contiguous()


def _cbr3d(m, **kw):
    """ConvBnReLU3D module -> ConvLayer."""
    return ConvLayer(m.conv.weight, False, m.conv.stride, m.conv.padding, bn=m.bn, relu=True, **kw)


def _up3d(seq, prob=None):
    """Sequential(ConvTranspose3d, BatchNorm3d, ReLU) -> ConvLayer."""
    ct, bn = seq[0], seq[1]
    return ConvLayer(ct.weight, True, ct.stride, ct.padding, bn=bn, relu=True, prob=prob)


def fused_conv11_select(L, t, c0, hypo, split_itv, inverse_depth, want_logits=False):
    """reg2d's conv11 (+ BatchNorm, ReLU, skip c0) + `prob` + softmax / argmax / gather / bounds in one launch
    (mvster_deconv_select): t [B,D,hi,wi,16], c0 [B,D,2hi,2wi,8], hypo [B,D,2hi,2wi] -> the stage's selection dict."""
    B, D, hi, wi, _ = t.shape
    dev = t.device
    hypo = hypo.contiguous()
    attn = torch.empty(B, D, 2 * hi, 2 * wi, device=dev, dtype=torch.float32)
    depth = torch.empty(B, 2 * hi, 2 * wi, device=dev, dtype=torch.float32)
    conf = torch.empty_like(depth)
    imin = torch.empty_like(depth) if inverse_depth else None
    imax = torch.empty_like(depth) if inverse_depth else None
    lo = torch.empty_like(attn) if want_logits else None
    rc = _lib.load().mvster_deconv_select(
        t.data_ptr(), L.w_deconv.data_ptr(), L.scale.data_ptr(), L.shift.data_ptr(), c0.data_ptr(),
        L.prob[0].data_ptr(), L.prob[1].data_ptr(), hypo.data_ptr(), attn.data_ptr(), depth.data_ptr(), conf.data_ptr(),
        None if imin is None else imin.data_ptr(), None if imax is None else imax.data_ptr(),
        None if lo is None else lo.data_ptr(), B, D, hi, wi, L.cin, int(L.relu), float(split_itv), ops._stream())
    _lib.check(rc, "deconv_select")
    out = {"attn_weight": attn, "depth": depth, "conf": conf}
    if inverse_depth:
        out["inverse_min_depth"], out["inverse_max_depth"] = imin, imax
    if want_logits:
        out["logits"] = lo
    return out


class Reg2dPlan:
    """reg2d U-Net (models/mvs4net_utils.py:870-912) on channels-last volumes; the 1x1x1 ``prob``
    head is left to the selection kernel (fused with the softmax)."""

    def __init__(self, m, fuse_prob_into_conv11=True):
        self.conv0, self.conv1, self.conv2 = _cbr3d(m.conv0), _cbr3d(m.conv1), _cbr3d(m.conv2)
        self.conv3, self.conv4 = _cbr3d(m.conv3), _cbr3d(m.conv4)
        self.conv5, self.conv6 = _cbr3d(m.conv5), _cbr3d(m.conv6)
        self.conv7, self.conv9 = _up3d(m.conv7), _up3d(m.conv9)
        self.prob_w = m.prob.weight.detach().float().reshape(-1).contiguous()
        self.prob_b = m.prob.bias.detach().float().reshape(-1).contiguous()
        # the 1x1x1 `prob` head runs in conv11's epilogue (the 8-channel volume is never written); with
        # fuse_prob_into_conv11=False the plan returns the feature volume and the selection kernel applies it
        self.conv11 = _up3d(m.conv11, prob=(self.prob_w, self.prob_b) if fuse_prob_into_conv11 else None)
        self.fused_prob = not fuse_prob_into_conv11

    def layers(self):
        return [self.conv0, self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.conv6, self.conv7,
                self.conv9, self.conv11]

    def __call__(self, x):
        """x [B,D,h,w,G] -> logits [B,D,h,w] (or the last feature volume [B,D,h,w,8] if ``fused_prob``)."""
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        t = self.conv6(self.conv5(c4))
        t = self.conv7(t, skip=c4, skip_mode=SKIP_ADD)
        t = self.conv9(t, skip=c2, skip_mode=SKIP_ADD)
        return self.conv11(t, skip=c0, skip_mode=SKIP_ADD)

    def select(self, x, hypo, split_itv, inverse_depth, want_logits=False):
        """x [B,D,h,w,G] -> the stage's selection dict (ops.select_depth's): the U-Net, then conv11 + `prob` + softmax +
        argmax + gather + bounds in ONE launch (mvster_deconv_select) when conv11 runs on the narrow deconvolution kernel;
        otherwise logits + ops.select_depth.  Same bits either way."""
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        t = self.conv6(self.conv5(c4))
        t = self.conv7(t, skip=c4, skip_mode=SKIP_ADD)
        t = self.conv9(t, skip=c2, skip_mode=SKIP_ADD)
        L = self.conv11
        B, D, hi, wi, _ = t.shape
        if (FUSE_SELECT and L.w_deconv is not None and L.prob is not None and L.cin == 16 and FORCE_VARIANT in (None, 4)
                and 2 <= D <= 16 and t.is_contiguous() and c0.is_contiguous()):
            return fused_conv11_select(L, t, c0, hypo, split_itv, inverse_depth, want_logits)
        res = self.conv11(t, skip=c0, skip_mode=SKIP_ADD)
        if self.fused_prob:
            return ops.select_depth(hypo, split_itv, inverse_depth, feat_cl=res, prob_w=self.prob_w, prob_b=self.prob_b,
                                    want_logits=want_logits)
        return ops.select_depth(hypo, split_itv, inverse_depth, logits=res, want_logits=want_logits)

    def flops(self, B, D, h, w):
        tot, shp = 0, None
        dims = {"conv0": (D, h, w), "conv1": (D, h, w), "conv2": (D, h // 2, w // 2), "conv3": (D, h // 2, w // 2),
                "conv4": (D, h // 4, w // 4), "conv5": (D, h // 4, w // 4), "conv6": (D, h // 8, w // 8),
                "conv7": (D, h // 8, w // 8), "conv9": (D, h // 4, w // 4), "conv11": (D, h // 2, w // 2)}
        for k, (d_, h_, w_) in dims.items():
            tot += getattr(self, k).flops(B, d_, h_, w_)
        return tot + 2 * B * D * h * w * 8


class Reg3dPlan:
    """reg3d U-Net (models/mvs4net_utils.py:914-965); ``prob`` is a 3x3x3 conv -> logits."""

    def __init__(self, m):
        self.down_size = m.down_size
        self.conv0, self.conv1, self.conv2 = _cbr3d(m.conv0), _cbr3d(m.conv1), _cbr3d(m.conv2)
        if m.down_size >= 2:
            self.conv3, self.conv4 = _cbr3d(m.conv3), _cbr3d(m.conv4)
            self.conv9 = _up3d(m.conv9)
        if m.down_size >= 3:
            self.conv5, self.conv6 = _cbr3d(m.conv5), _cbr3d(m.conv6)
            self.conv7 = _up3d(m.conv7)
        self.conv11 = _up3d(m.conv11)
        self.prob = ConvLayer(m.prob.weight, False, m.prob.stride, m.prob.padding)
        self.fused_prob = False

    def __call__(self, x):
        """x [B,D,h,w,G] -> logits [B,D,h,w]."""
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        if self.down_size == 3:
            c4 = self.conv4(self.conv3(c2))
            t = self.conv6(self.conv5(c4))
            t = self.conv7(t, skip=c4, skip_mode=SKIP_ADD)
            t = self.conv9(t, skip=c2, skip_mode=SKIP_ADD)
        elif self.down_size == 2:
            t = self.conv4(self.conv3(c2))
            t = self.conv9(t, skip=c2, skip_mode=SKIP_ADD)
        else:
            t = c2
        t = self.conv11(t, skip=c0, skip_mode=SKIP_ADD)
        return self.prob(t).squeeze(-1)


def _cbr2d(m, **kw):
    c = m.conv
    return ConvLayer(c.weight, False, c.stride, c.padding, bn=m.bn, relu=m.relu, **kw)


def _plain2d(c):
    return ConvLayer(c.weight, False, c.stride, c.padding, bias=c.bias)


class FpnPlan:
    """FPN4 (models/mvs4net_utils.py:419-502) for all views at once; input [N*B,1,H,W,4] (RGB0),
    outputs four channels-last maps [N*B,1,h,w,C]."""

    def __init__(self, m):
        self.conv0 = [_cbr2d(m.conv0[0], cin_pad=4), _cbr2d(m.conv0[1])]
        self.conv1 = [_cbr2d(l) for l in m.conv1]
        self.conv2 = [_cbr2d(l) for l in m.conv2]
        self.conv3 = [_cbr2d(l) for l in m.conv3]
        self.inner1 = _plain2d(m.inner1)
        self.out1, self.out2 = _plain2d(m.out1), _plain2d(m.out2)
        # The two fine levels, re-associated so that neither top-down map
        #   f2 = F.interpolate(f1) + inner2(c1)            (105 MB at 5 x 512 x 640, half resolution)
        #   f3 = F.interpolate(f2) + inner3(c0)            (419 MB, full resolution)
        # is ever formed.  The output convs are linear and interpolation acts per channel, hence for level L
        #   outL(f)[p] = sum_tap up(WL[tap] f_coarser)[p+tap] + sum_tap WL[tap] (Wi c[p+tap] + bi)
        # = gather-sum of G = (1x1 conv 64 -> 9*co of the coarser map, at ITS resolution)  [*_g + fpn_tail_gather]
        # + 3x3 conv of the bottom-up map c with composed weights WL[tap] @ Wi               [*_c, skip-add]
        # + the bias pushed through the in-bounds taps                                        [*_vb]
        # Level 4 needs G4 = Wg4 f2 at half resolution; f2 itself is a sum, so
        #   G4 = up(Wg4 f1) + (Wg4 W2) c1 + Wg4 b2         [tail_g at QUARTER resolution + fpn_lateral_up]
        w2 = m.inner2.weight.detach().double()[:, :, 0, 0]      # [64, 16]
        b2 = m.inner2.bias.detach().double()                    # [64]
        self.mid_g, self.mid_c, self.mid_vb, _ = self._reassociate(m.out3, w2, b2)
        w3 = m.inner3.weight.detach().double()[:, :, 0, 0]      # [64, 8]
        b3 = m.inner3.bias.detach().double()                    # [64]
        self.tail_g, self.tail_c, self.tail_vb, wg4 = self._reassociate(m.out4, w3, b3)      # wg4 [72, 64], fp64
        self.tail_a = (wg4 @ w2).float().contiguous()           # [72, 16]   [ops.fpn_lateral_up]
        self.tail_ab = (wg4 @ b2).float().contiguous()

    @staticmethod
    def _reassociate(out, wi, bi):
        """out: the level's 3x3 output conv; wi [64, cin], bi [64]: the lateral 1x1 conv -> (g, c, vb, g's fp64 matrix)."""
        wo = out.weight.detach().double()                                        # [co, 64, 3, 3]
        co = wo.shape[0]
        wg = wo.permute(2, 3, 0, 1).reshape(9 * co, wo.shape[1])                 # row = tap*co + co_idx
        wc = torch.einsum("ocyx,ci->oiyx", wo, wi)                               # [co, cin, 3, 3]
        g = ConvLayer(wg.float().reshape(9 * co, wo.shape[1], 1, 1), False, (1, 1), (0, 0))
        c = ConvLayer(wc.float(), False, (1, 1), (1, 1), bias=out.bias)
        vb = torch.einsum("ocyx,c->yxo", wo, bi).reshape(9, co).float().contiguous()
        return g, c, vb, wg

    @staticmethod
    def _seq(layers, x):
        for l in layers:
            x = l(x)
        return x

    def trunk(self, x):
        """Bottom-up path + the first top-down map f1: what both branches below need."""
        c0 = self._conv0(x)
        c1 = self._seq(self.conv1, c0)
        c2 = self._seq(self.conv2, c1)
        c3 = self._seq(self.conv3, c2)
        f1 = self.inner1(c2, skip=c3, skip_mode=SKIP_UPSAMPLE_ADD)
        return c0, c1, c3, f1

    def _conv0(self, x):
        """conv0[0] -> conv0[1] (3 -> 8 -> 8 at full resolution) in one launch, the intermediate in LDS (conv_narrow_pair_kernel:
        78 instead of 183 MB of HBM traffic at 5 x 512 x 640); small maps and forced variants take the two layers' own launches."""
        a, b = self.conv0
        N, D, H, W, _ = x.shape
        if (FUSE_CONV0 and FORCE_VARIANT is None and a.w_small is not None and b.w_small is not None and a.cin == 4 and b.cin == 8
                and D == 1 and N * H * W >= NARROW_PAIR_MIN_PIXELS and x.is_cuda and a.w_small.device == x.device and x.is_contiguous() and x.dtype == torch.float32
                and hasattr(_lib.load(), "mvster_conv_narrow_pair")):
            out = torch.empty((N, 1, H, W, 8), device=x.device, dtype=torch.float32)
            rc = _lib.load().mvster_conv_narrow_pair(
                x.data_ptr(), a.w_small.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr(), b.w_small.data_ptr(),
                b.scale.data_ptr(), b.shift.data_ptr(), out.data_ptr(), N, H, W, int(a.relu), int(b.relu), NARROW_PAIR_WPC, ops._stream())
            _lib.check(rc, "conv_narrow_pair")
            return out
        return self._seq(self.conv0, x)

    def coarse(self, c3, f1):
        """Output convs of the two coarse levels: everything stages 1 and 2 need."""
        return self.out1(c3), self.out2(f1)

    def head(self, x):
        c0, c1, c3, f1 = self.trunk(x)
        o1, o2 = self.coarse(c3, f1)
        return c0, c1, f1, o1, o2

    def tail(self, c0, c1, f1):
        """The two fine levels (needed from stage 3 on); independent of the coarse outputs and of cascade
        stages 1-2, so the model runs it on a second HIP stream underneath them."""
        return self.tail_mid(c0, c1, f1), self.tail_fine(c0, c1, f1)

    def tail_mid(self, c0, c1, f1):
        """Level 3 (half resolution): what cascade stage 3 reads."""
        H, W = c0.shape[2], c0.shape[3]
        p3 = ops.fpn_tail_gather(self.mid_g(f1), self.mid_vb, H // 2, W // 2)
        return self.mid_c(c1, skip=p3, skip_mode=SKIP_ADD)

    def tail_fine(self, c0, c1, f1):
        """Level 4 (full resolution): what cascade stage 4 reads."""
        H, W = c0.shape[2], c0.shape[3]
        q4 = self.tail_g(f1)
        p4 = ops.fpn_tail_fused(c1, self.tail_a, self.tail_ab, q4, self.tail_vb, H, W) if FUSE_TAIL else None
        if p4 is None:                                                       # (small / odd maps: two launches)
            p4 = ops.fpn_tail_gather(ops.fpn_lateral_up(c1, self.tail_a, self.tail_ab, q4), self.tail_vb, H, W)
        return self.tail_c(c0, skip=p4, skip_mode=SKIP_ADD)

    def __call__(self, x):
        c0, c1, f1, o1, o2 = self.head(x)
        o3, o4 = self.tail(c0, c1, f1)
        return [o1, o2, o3, o4]
