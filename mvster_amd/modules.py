"""Parameter containers of the MVS4net module tree.

These modules own the learnable state with exactly the reference's ``state_dict`` keys
(348 tensors for the shipped configuration: ``feature.*`` 76, ``reg.{0..3}.*`` 62 each,
``mono_depth_decoder.*`` 24 -- SURVEY.md section 5), so reference checkpoints load with
``strict=True``.  ``forward_cl`` is the differentiable channels-last form on the gfx950 kernels
(``mvster_amd.train_ops``: conv -> BatchNorm -> ReLU, all passes hand-written HIP) that
``MVS4net`` runs in training mode, where BatchNorm works on batch statistics and cannot be folded;
``forward`` is the same thing behind the reference's NCHW / NCDHW signatures (a permute on either
side, no second backend).  In eval mode ``mvster_amd.net.MVS4net`` bypasses both and runs the
folded-BatchNorm plans of ``mvster_amd.conv_plan``.

Reference: models/mvs4net_utils.py:116-123 (ConvBnReLU3D), :224-251 (Conv2d), :419-502 (FPN4),
:833-868 (mono_depth_decoder), :870-912 (reg2d), :914-965 (reg3d).
"""
import contextlib

import torch
import torch.nn as nn

from . import train_ops as T


def _to_cl5(x):      # [B,C,D,H,W] -> [B,D,H,W,C]
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _from_cl5(y):    # [B,D,H,W,C] -> [B,C,D,H,W] view
    return y.permute(0, 4, 1, 2, 3)


def _to_cl4(x):      # [B,C,H,W] -> [B,1,H,W,C]
    return x.permute(0, 2, 3, 1).unsqueeze(1).contiguous()


def _from_cl4(y):    # [B,1,H,W,C] -> [B,C,H,W] view
    return y[:, 0].permute(0, 3, 1, 2)


class ConvBnReLU3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(out_channels)

    def forward(self, x):
        return _from_cl5(self.forward_cl(_to_cl5(x)))

    def forward_cl(self, x, tap=False):
        """Channels-last [B,D,H,W,C] form on the gfx950 kernels (mvster_amd/train_ops.py).  ``tap``: -> (y, x') with x' the
        alias of x that x's other consumers should read (train_ops.conv_cl)."""
        c = self.conv
        if tap:
            y, xt = T.conv_cl(x, c.weight, None, c.stride, c.padding, tap=True)
            return T.batch_norm_cl(y, self.bn, relu=True), xt
        return T.batch_norm_cl(T.conv_cl(x, c.weight, None, c.stride, c.padding), self.bn, relu=True)


def _deconv_bn_relu(cin, cout, kernel, pad, out_pad, stride):
    return nn.Sequential(
        nn.ConvTranspose3d(cin, cout, kernel_size=kernel, padding=pad, output_padding=out_pad, stride=stride, bias=False),
        nn.BatchNorm3d(cout), nn.ReLU(inplace=True))


def _deconv_bn_relu_cl(seq, x, skip=None):
    """ConvTranspose3d -> BatchNorm -> ReLU (+ the U-Net skip connection, fused into the BatchNorm apply kernel)."""
    ct, bn = seq[0], seq[1]
    return T.batch_norm_cl(T.conv_cl(x, ct.weight, None, ct.stride, ct.padding, transposed=True), bn, relu=True, skip=skip)


class reg2d(nn.Module):
    def __init__(self, input_channel=128, base_channel=32, conv_name="ConvBnReLU3D"):
        super().__init__()
        if conv_name != "ConvBnReLU3D":
            raise NotImplementedError("agg_type %r: only the shipped ConvBnReLU3D blocks are built" % conv_name)
        c = base_channel
        flat = dict(kernel_size=(1, 3, 3), pad=(0, 1, 1))
        down = dict(kernel_size=(1, 3, 3), stride=(1, 2, 2), pad=(0, 1, 1))
        self.conv0 = ConvBnReLU3D(input_channel, c, **flat)
        self.conv1 = ConvBnReLU3D(c, c * 2, **down)
        self.conv2 = ConvBnReLU3D(c * 2, c * 2)
        self.conv3 = ConvBnReLU3D(c * 2, c * 4, **down)
        self.conv4 = ConvBnReLU3D(c * 4, c * 4)
        self.conv5 = ConvBnReLU3D(c * 4, c * 8, **down)
        self.conv6 = ConvBnReLU3D(c * 8, c * 8)
        up = ((1, 3, 3), (0, 1, 1), (0, 1, 1), (1, 2, 2))
        self.conv7 = _deconv_bn_relu(c * 8, c * 4, *up)
        self.conv9 = _deconv_bn_relu(c * 4, c * 2, *up)
        self.conv11 = _deconv_bn_relu(c * 2, c, *up)
        self.prob = nn.Conv3d(8, 1, 1, stride=1, padding=0)   # hard-coded 8 inputs, with bias (reference :900)

    def forward(self, x):
        """[B,G,D,h,w] -> logits [B,D,h,w] (the reference's signature)."""
        return self.forward_cl(_to_cl5(x))

    def forward_cl(self, x, return_features=False):
        """[B,D,h,w,G] -> logits [B,D,h,w]; ``return_features``: the 8-channel volume in front of the ``prob`` head instead
        (MVS4net's training forward applies the head inside its fused selection kernel)."""
        # (c0 / c2 / c4 feed the next level AND a skip connection: the skip reads the tap, so that its gradient is added
        #  inside the strided layer's input-gradient kernel)
        c0 = self.conv0.forward_cl(x)
        x, c0 = self.conv1.forward_cl(c0, tap=True)
        c2 = self.conv2.forward_cl(x)
        x, c2 = self.conv3.forward_cl(c2, tap=True)
        c4 = self.conv4.forward_cl(x)
        x, c4 = self.conv5.forward_cl(c4, tap=True)
        x = self.conv6.forward_cl(x)
        x = _deconv_bn_relu_cl(self.conv7, x, skip=c4)
        x = _deconv_bn_relu_cl(self.conv9, x, skip=c2)
        x = _deconv_bn_relu_cl(self.conv11, x, skip=c0)
        if return_features:
            return x
        # 1x1x1 conv 8 -> 1 as multiply + 8-wide reduction (rocBLAS gemv takes 8 ms on this [2.6 M, 8] shape)
        return (x * self.prob.weight.reshape(-1)).sum(-1) + self.prob.bias


class reg3d(nn.Module):
    def __init__(self, in_channels, base_channels, down_size=3):
        super().__init__()
        c = base_channels
        self.down_size = down_size
        self.conv0 = ConvBnReLU3D(in_channels, c, kernel_size=3, pad=1)
        self.conv1 = ConvBnReLU3D(c, c * 2, kernel_size=3, stride=2, pad=1)
        self.conv2 = ConvBnReLU3D(c * 2, c * 2)
        if down_size >= 2:
            self.conv3 = ConvBnReLU3D(c * 2, c * 4, kernel_size=3, stride=2, pad=1)
            self.conv4 = ConvBnReLU3D(c * 4, c * 4)
        if down_size >= 3:
            self.conv5 = ConvBnReLU3D(c * 4, c * 8, kernel_size=3, stride=2, pad=1)
            self.conv6 = ConvBnReLU3D(c * 8, c * 8)
            self.conv7 = _deconv_bn_relu(c * 8, c * 4, 3, 1, 1, 2)
        if down_size >= 2:
            self.conv9 = _deconv_bn_relu(c * 4, c * 2, 3, 1, 1, 2)
        self.conv11 = _deconv_bn_relu(c * 2, c, 3, 1, 1, 2)
        self.prob = nn.Conv3d(c, 1, 3, stride=1, padding=1, bias=False)

    def forward(self, x):
        return self.forward_cl(_to_cl5(x))

    def forward_cl(self, x):
        c0 = self.conv0.forward_cl(x)
        x, c0 = self.conv1.forward_cl(c0, tap=True)
        c2 = self.conv2.forward_cl(x)
        if self.down_size == 3:
            x, c2 = self.conv3.forward_cl(c2, tap=True)
            c4 = self.conv4.forward_cl(x)
            x, c4 = self.conv5.forward_cl(c4, tap=True)
            x = self.conv6.forward_cl(x)
            x = _deconv_bn_relu_cl(self.conv7, x, skip=c4)
            x = _deconv_bn_relu_cl(self.conv9, x, skip=c2)
        elif self.down_size == 2:
            x, c2 = self.conv3.forward_cl(c2, tap=True)
            x = self.conv4.forward_cl(x)
            x = _deconv_bn_relu_cl(self.conv9, x, skip=c2)
        else:
            x = c2
        x = _deconv_bn_relu_cl(self.conv11, x, skip=c0)
        p = self.prob
        return T.conv_cl(x, p.weight, None, p.stride, p.padding).squeeze(-1)


class Conv2d(nn.Module):
    """conv (no bias) + BatchNorm2d + optional ReLU (reference Conv2d with gn=False)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn_momentum=0.1, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=False, **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum)
        self.relu = relu

    def forward(self, x):
        return _from_cl4(self.forward_cl(_to_cl4(x)))

    def forward_cl(self, x, groups=1, tap=False):
        """[B,1,H,W,C] channels-last; ``groups`` = number of equal batch slices normalised separately (views); ``tap``: ->
        (y, x') with x' the alias of x that x's other consumers should read (train_ops.conv_cl)."""
        c = self.conv
        if tap:
            y, xt = T.conv_cl(x, c.weight, None, c.stride, c.padding, tap=True)
            return T.batch_norm_cl(y, self.bn, relu=self.relu, groups=groups), xt
        return T.batch_norm_cl(T.conv_cl(x, c.weight, None, c.stride, c.padding), self.bn, relu=self.relu, groups=groups)


class FPN4(nn.Module):
    def __init__(self, base_channels, gn=False, dcn=False):
        super().__init__()
        if gn or dcn:
            raise NotImplementedError("FPN4: GroupNorm / deformable-conv variants are out of scope (SURVEY.md section 2, #10)")
        c = base_channels
        self.base_channels = c
        self.conv0 = nn.Sequential(Conv2d(3, c, 3, 1, padding=1), Conv2d(c, c, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(c, c * 2, 5, stride=2, padding=2), Conv2d(c * 2, c * 2, 3, 1, padding=1),
                                   Conv2d(c * 2, c * 2, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(c * 2, c * 4, 5, stride=2, padding=2), Conv2d(c * 4, c * 4, 3, 1, padding=1),
                                   Conv2d(c * 4, c * 4, 3, 1, padding=1))
        self.conv3 = nn.Sequential(Conv2d(c * 4, c * 8, 5, stride=2, padding=2), Conv2d(c * 8, c * 8, 3, 1, padding=1),
                                   Conv2d(c * 8, c * 8, 3, 1, padding=1))
        f = c * 8
        self.inner1 = nn.Conv2d(c * 4, f, 1, bias=True)
        self.inner2 = nn.Conv2d(c * 2, f, 1, bias=True)
        self.inner3 = nn.Conv2d(c, f, 1, bias=True)
        self.out1 = nn.Conv2d(f, c * 8, 1, bias=False)
        self.out2 = nn.Conv2d(f, c * 4, 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(f, c * 2, 3, padding=1, bias=False)
        self.out4 = nn.Conv2d(f, c, 3, padding=1, bias=False)
        self.out_channels = [c * 8, c * 4, c * 2, c]
        self.tail_stream = None        # a HIP stream for the two fine levels' forward and backward (set per call by MVS4net)
        self.tail_pending = None       # ... the stream the last forward_cl left un-joined, for the caller to wait on

    def forward(self, x):
        """[B,3,H,W] -> {"stage1".."stage4": [B,C,h,w]} (the reference's signature; views of channels-last maps)."""
        return {k: _from_cl4(v) for k, v in self.forward_cl(_to_cl4(x)).items()}

    def forward_cl(self, x, groups=1):
        """x [B,1,H,W,3] channels-last -> four channels-last maps [B,1,h,w,C].  With ``groups`` = N the batch holds
        the N views of every sample, view-major ([N*B,...]): the convolutions run once over all of them, BatchNorm
        normalises each view's slice on its own -- the same numbers as N separate calls (the reference calls the
        FPN once per view, MVS4Net.py:65-68), in a fifth of the launches."""
        def seq(layers, t, tap=False):
            # (`tap`: -> (y, t') with t' the alias of the input for its second consumer, see train_ops.conv_cl)
            t0 = None
            for i, l in enumerate(layers):
                if tap and i == 0:
                    t, t0 = l.forward_cl(t, groups, tap=True)
                else:
                    t = l.forward_cl(t, groups)
            return (t, t0) if tap else t

        def plain(m, t, up=None, tap=False):
            # (`up`: the coarser level, up-sampled x2 and added inside the convolution's epilogue)
            return T.conv_cl(t, m.weight, m.bias, m.stride, m.padding, skip=up, skip_upsample=up is not None, tap=tap)
        # every level's map has two consumers (the next level / output conv and a lateral): the second one reads the tap
        c0 = seq(self.conv0, x)
        c1, c0 = seq(self.conv1, c0, tap=True)
        c2, c1 = seq(self.conv2, c1, tap=True)
        c3, c2 = seq(self.conv3, c2, tap=True)
        out = {}
        out["stage1"], c3 = plain(self.out1, c3, tap=True)
        f = plain(self.inner1, c2, up=c3)
        out["stage2"], f = plain(self.out2, f, tap=True)
        # The two fine levels on `tail_stream` when the caller set one (MVS4net._forward_train): nothing before cascade stage 3
        # reads them, so they run beside stages 1 and 2 (small, latency-bound launches); the caller joins the stream before
        # stage 3 (`tail_pending`).  Autograd runs their backward on the same stream.
        side = self.tail_stream if x.is_cuda else None
        self.tail_pending = side
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(x.device))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            f = plain(self.inner2, c1, up=f)
            out["stage3"], f = plain(self.out3, f, tap=True)
            # finest level: re-associated, the full-resolution 64-channel map is never formed (train_ops.fpn_fine_level)
            # (autograd reaches this output when every cascade stage's backward has been issued and before the FPN's begins:
            #  the point where a deferred_wgrad_finish(early=True) launches the weight gradients collected so far)
            out["stage4"] = T.wgrad_flush_point(T.fpn_fine_level(c0, f, self.inner3, self.out4))
        return out


class mono_depth_decoder(nn.Module):
    """Training-only auxiliary monocular head."""

    def __init__(self):
        super().__init__()
        self.convblocks = nn.ModuleList([Conv2d(64, 32, 3, 1, padding=1), Conv2d(32, 16, 3, 1, padding=1),
                                         Conv2d(16, 8, 3, 1, padding=1)])
        self.conv3x3 = nn.ModuleList([nn.Conv2d(64, 1, 3, 1, 1), nn.Conv2d(32, 1, 3, 1, 1), nn.Conv2d(16, 1, 3, 1, 1)])

    def forward(self, outputs, d_min, d_max):
        """The reference's signature: reads ``outputs[stage]["mono_feat"]`` ([B,C,h,w])."""
        return self.forward_cl(outputs, [_to_cl4(outputs["stage%d" % i]["mono_feat"]) for i in range(1, 5)], d_min, d_max)

    def forward_cl(self, outputs, feats_cl, d_min, d_max):
        """Same head on the channels-last reference features ``feats_cl`` [stage] -> [B,1,h,w,C]."""
        d_min, d_max = d_min.detach().float().contiguous(), d_max.detach().float().contiguous()
        # (stages 2 and 3 feed a conv block AND the next concatenation: the latter reads the block's tap)
        blocks, feats = [], list(feats_cl)
        for j in range(3):
            if j == 0:
                blocks.append(self.convblocks[j].forward_cl(feats[j]))
            else:
                y, feats[j] = self.convblocks[j].forward_cl(feats[j], tap=True)
                blocks.append(y)
        for i in range(1, 4):
            c = self.conv3x3[i - 1]
            # nearest x2 + concatenation in one launch; sigmoid -> disparity range -> reciprocal in one launch
            z = T.conv_cl(T.upcat_cl(blocks[i - 1], feats[i]), c.weight, c.bias, c.stride, c.padding)
            outputs["stage%d" % (i + 1)]["mono_depth"] = T.mono_depth_cl(z.view(z.shape[0], z.shape[2], z.shape[3]), d_min, d_max)
        return outputs
