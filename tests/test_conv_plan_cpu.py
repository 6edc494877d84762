"""conv_plan (weight packing, BN folding, geometry, parity classes) checked on CPU through a
NumPy emulation of the HIP kernel's indexing, against the oracle's PyTorch modules."""
import numpy as np
import pytest
import torch

from mvster_amd import conv_plan as cp
from mvster_amd import modules as M
from mvster_amd.synthetic import randomize_state
from oracle import mvs4_oracle as O
from tests.conv_emulator import Emulated, run_layer


def cl(x):   # NCDHW -> NDHWC
    return x.permute(0, 2, 3, 4, 1).contiguous()


def twin(oracle_module, m):
    """The oracle's PyTorch module with the product module's parameters (same state_dict keys by construction):
    the CPU reference of every check here.  The product modules themselves run on the GPU only."""
    oracle_module.load_state_dict(m.state_dict(), strict=True)
    return oracle_module.eval()


def test_pack_gemm_layout():
    wk = torch.arange(40 * 24, dtype=torch.float32).reshape(40, 24)
    flat, nsteps = cp._pack_gemm(wk)
    assert nsteps == 3 and flat.numel() == 48 * 32
    f = flat.reshape(3, 2, 64, 4)
    for s, t, lane, j in [(0, 0, 0, 0), (1, 1, 37, 2), (2, 0, 63, 3), (2, 1, 5, 1)]:
        k, n = s * 16 + (lane >> 4) * 4 + j, t * 16 + (lane & 15)
        want = wk[k, n].item() if (k < 40 and n < 24) else 0.0
        assert f[s, t, lane, j].item() == want


@pytest.mark.parametrize("cin,cout,k,s,p", [(8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)), (4, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                            (8, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1)), (16, 16, 3, 1, 1),
                                            (32, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)), (16, 32, 3, 2, 1)])
def test_conv_bn_relu_layer(cin, cout, k, s, p):
    torch.manual_seed(cin * 100 + cout)
    m = M.ConvBnReLU3D(cin, cout, kernel_size=k, stride=s, pad=p)
    m.load_state_dict(randomize_state(m.state_dict(), seed=3))
    m.eval()
    x = torch.randn(2, cin, 4, 10, 12)
    with torch.no_grad():
        want = twin(O._CBR3d(cin, cout, kernel_size=k, stride=s, pad=p), m)(x)
    got = run_layer(cp._cbr3d(m), cl(x))
    assert got.shape == cl(want).shape
    assert (got - cl(want)).abs().max() <= 2e-5 * want.abs().max()


@pytest.mark.parametrize("cin,cout,k,pad,op,s", [(64, 32, (1, 3, 3), (0, 1, 1), (0, 1, 1), (1, 2, 2)),
                                                 (16, 8, (1, 3, 3), (0, 1, 1), (0, 1, 1), (1, 2, 2)),
                                                 (16, 8, 3, 1, 1, 2)])
def test_transposed_layer_with_skip(cin, cout, k, pad, op, s):
    torch.manual_seed(cin + cout)
    seq = M._deconv_bn_relu(cin, cout, k, pad, op, s)
    seq.load_state_dict(randomize_state(seq.state_dict(), seed=4))
    seq.eval()
    x = torch.randn(2, cin, 3, 5, 6)
    with torch.no_grad():
        y = twin(O._up3d(cin, cout, k, pad, op, s), seq)(x)
        skip = torch.randn_like(y)
        want = skip + y
    got = run_layer(cp._up3d(seq), cl(x), skip=cl(skip), skip_mode=cp.SKIP_ADD)
    assert (got - cl(want)).abs().max() <= 2e-5 * want.abs().max()


@pytest.mark.parametrize("G,D", [(8, 8), (4, 4)])
def test_reg2d_plan(G, D):
    torch.manual_seed(G)
    m = M.reg2d(input_channel=G, base_channel=8)
    m.load_state_dict(randomize_state(m.state_dict(), seed=5, prob_gain=1.0))
    m.eval()
    x = torch.randn(1, G, D, 16, 24)
    with torch.no_grad():
        want = twin(O.Reg2d(input_channel=G, base_channel=8), m)(x)
    plan = cp.Reg2dPlan(m, fuse_prob_into_conv11=False)
    feat = Emulated(plan)(cl(x))                                 # [B,D,h,w,8]
    logits = feat @ plan.prob_w + plan.prob_b
    assert (logits - want).abs().max() <= 5e-5 * want.abs().max()
    fused = Emulated(cp.Reg2dPlan(m))(cl(x))                     # prob head in conv11's epilogue -> [B,D,h,w]
    assert fused.shape == want.shape and (fused - want).abs().max() <= 5e-5 * want.abs().max()


@pytest.mark.parametrize("down", [3, 2, 1])
def test_reg3d_plan(down):
    torch.manual_seed(down)
    m = M.reg3d(in_channels=8, base_channels=8, down_size=down)
    m.load_state_dict(randomize_state(m.state_dict(), seed=6, prob_gain=1.0))
    m.eval()
    x = torch.randn(1, 8, 8, 8, 16)
    with torch.no_grad():
        want = twin(O.Reg3d(in_channels=8, base_channels=8, down_size=down), m)(x)
    got = Emulated(cp.Reg3dPlan(m))(cl(x))
    assert (got - want).abs().max() <= 5e-5 * want.abs().max()


def test_fpn_plan():
    torch.manual_seed(0)
    m = M.FPN4(base_channels=8)
    m.load_state_dict(randomize_state(m.state_dict(), seed=7))
    m.eval()
    img = torch.rand(2, 3, 32, 48)
    with torch.no_grad():
        want = twin(O.FPN4(base_channels=8), m)(img)
    x = torch.zeros(2, 1, 32, 48, 4)
    x[:, 0, :, :, :3] = img.permute(0, 2, 3, 1)
    got = Emulated(cp.FpnPlan(m))(x)
    for s in range(4):
        w = want["stage%d" % (s + 1)].permute(0, 2, 3, 1)
        assert got[s][:, 0].shape == w.shape
        assert (got[s][:, 0] - w).abs().max() <= 5e-5 * w.abs().max(), s


def test_tile_heuristic_and_flops():
    assert cp._tiles(4 * 512 * 640, 1, 1) == (4, 1)
    mt, nt = cp._tiles(8 * 8 * 10, 4, 1)
    assert (mt, nt) == (1, 1)
    m = M.reg2d(input_channel=4, base_channel=8).eval()
    plan = cp.Reg2dPlan(m)
    # 14,416 conv FLOPs per D*h*w voxel for G=4
    assert plan.flops(1, 4, 64, 64) == 14416 * 4 * 64 * 64
    m8 = M.reg2d(input_channel=8, base_channel=8).eval()
    assert cp.Reg2dPlan(m8).flops(1, 8, 64, 64) == 14992 * 8 * 64 * 64


def test_tuning_table_is_well_formed():
    """mvster_amd/tuning_gfx950.json (signature -> [variant word, mt, nt], scripts/conv_microbench.py --emit): every key is
    a layer signature, every variant word names a kernel family that covers that layer (conv_mfma.hip routes on the low
    byte, the families read their mode from the bits above), and the tile counts are ones the launchers instantiate.  An
    entry outside these sets would turn into MVSTER_ERR_UNSUPPORTED on the first forward of that shape."""
    import json
    import os
    import re
    with open(os.path.join(os.path.dirname(cp.__file__), "tuning_gfx950.json")) as f:
        table = json.load(f)
    assert len(table) > 300
    key = re.compile(r"^([CT])(\d+)-(\d+)_k(\d)x(\d)x(\d)_s(\d)x(\d)x(\d)_(\d+)x(\d+)x(\d+)x(\d+)_sk([012])$")
    modes = {0: {0}, 1: {0}, 2: {0}, 3: {0}, 5: {1, 2, 3, 33, 34}, 6: {2, 3}, 7: {0, 1, 2}, 8: {1, 2}, 9: {0, 1}, 10: {0}}
    for sig, val in table.items():
        m = key.match(sig)
        assert m, sig
        assert isinstance(val, list) and len(val) == 3 and all(isinstance(v, int) for v in val), sig
        variant, mt, nt = val
        fam, mode = variant & 0xff, variant >> 8
        assert fam in modes and mode in modes[fam], (sig, val)
        if fam in (3, 10):      # narrow layers: VALU kernel (no tile arguments) or the shift-packed MFMA kernel (mt = tile rows / 4,
            #                     nt = workgroups per CU)
            assert (mt, nt) == (0, 0) if fam == 3 else (mt in (2, 4) and nt in (1, 2)), (sig, val)
            assert sig.startswith(("C4-8_k1x3x3_s1x1x1_", "C8-8_k1x3x3_s1x1x1_")), sig
            continue
        assert mt in (1, 2, 4) and nt in (1, 2, 3, 4, 5), (sig, val)
        transposed, cin, cout = m.group(1) == "T", int(m.group(2)), int(m.group(3))
        kernel = tuple(int(m.group(i)) for i in (4, 5, 6))
        stride = tuple(int(m.group(i)) for i in (7, 8, 9))
        skip = int(m.group(14))
        assert cin in (4, 8, 16, 32, 64, 80) and cout <= 144, sig         # (80: the padded 72-channel gradient of the FPN gather)
        if fam in (8, 9):       # Winograd F(2x2, 3x3): ConvLayer.wino_eligible(), no up-sampling skip
            assert not transposed and kernel in ((1, 3, 3), (3, 3, 3)) and stride == (1, 1, 1), sig
            assert cin in (16, 32, 64) and cout % 16 == 0 and skip in (0, 1), sig
            if fam == 9:        # ring kernel: two N tiles per workgroup in modes 1 and 2 (word 9 | 1 << 8), one in mode 0
                assert nt == (2 if mode == 1 else 1), (sig, val)
        if fam == 6:            # persistent 1x1
            assert kernel == (1, 1, 1) and stride == (1, 1, 1) and not transposed, sig
        if fam == 5:            # persistent LDS-DMA frame: stride-1 16 -> 16, the stride-2 families, the transposed 3x3 layers
            assert kernel[1] == kernel[2] and kernel[1] in (3, 5), sig
            assert (mode & 32) == 0 or stride == (1, 2, 2), sig        # loading waves exist for the stride-2 instances only
        if skip == 2:           # bilinear x2 up-sampling add in the epilogue: the direct / split-K / persistent 1x1 kernels
            assert fam in (0, 1, 2, 6) and kernel == (1, 1, 1), sig


def test_family_fallback_of_the_tuning_table():
    """conv_plan.tuned_choice: exact entries win; a shape the table does not hold takes the choice of its layer family's entry
    nearest in log2(voxels) (within FAMILY_REACH octaves); unknown families and far-away sizes fall to the heuristics."""
    class L:
        transposed, cin, cout, kernel, stride = False, 16, 16, (1, 3, 3), (1, 1, 1)
    exact, how = cp.tuned_choice(L, 5, 1, 256, 320, 0)
    assert how == "exact" and exact == cp._tuning()["C16-16_k1x3x3_s1x1x1_5x1x256x320_sk0"]
    # 832 x 1152 x 5 (the reference's real "mid" workload): half-resolution FPN level, not in the table
    fam, how = cp.tuned_choice(L, 5, 1, 416, 576, 0)
    assert how == "family" and fam == cp._tuning()["C16-16_k1x3x3_s1x1x1_5x1x576x800_sk0"]     # 1.2 M voxels -> the 2.3 M entry (0.9 octaves; 5 x 256 x 320 is 1.5 away)
    assert cp.tuned_choice(L, 1, 1, 4, 4, 0) == (None, None)                                  # 16 voxels: nothing within reach
    L.cin, L.cout = 16, 48
    assert cp.tuned_choice(L, 5, 1, 256, 320, 0) == (None, None)                              # a family the table never saw
    # the family choice is a copy (callers unpack / edit it) and every family list is sorted by size
    fam[0] = -1
    assert cp.tuned_choice(L, 5, 1, 256, 320, 0) == (None, None)
    # depth counts: a 4-slice volume takes a 4-slice entry's choice even when an 8-slice entry is nearer in voxels
    saved = cp._TUNING, cp._FAMILIES
    try:
        cp._TUNING = {"C64-64_k3x3x3_s1x1x1_1x8x32x48_sk0": [265, 2, 2], "C64-64_k3x3x3_s1x1x1_1x4x64x80_sk0": [9, 2, 1]}
        cp._FAMILIES = None
        L.cin, L.cout, L.kernel = 64, 64, (3, 3, 3)
        assert cp.tuned_choice(L, 1, 4, 48, 64, 0) == ([9, 2, 1], "family")          # 12 288 voxels: equal to the 8-slice entry's
        assert cp.tuned_choice(L, 1, 8, 32, 48, 0) == ([265, 2, 2], "exact")
        assert cp.tuned_choice(L, 1, 8, 24, 32, 0) == ([265, 2, 2], "family")
    finally:
        cp._TUNING, cp._FAMILIES = saved


def test_winograd_plan_for_untuned_shapes():
    """_wino_plan: the heuristic behind shapes the table does not know -- a Winograd kernel wherever the map gives most CUs
    a work unit, nothing otherwise (the caller keeps the direct / split-K choice)."""
    class L:
        kernel = (1, 3, 3)
        cin = 16
        ntile_total = 1
    assert cp._wino_plan(L, 10, 1, 256, 320) == (8 | (1 << 8), 2, 1)
    assert cp._wino_plan(L, 1, 1, 64, 80) is None                       # 20 tiles
    L.cin, L.ntile_total = 64, 4
    v = cp._wino_plan(L, 5, 1, 128, 160)
    assert v == (9, 2, 2)                                               # 400 tiles x 2 pairs of N tiles
    assert cp._wino_plan(L, 1, 1, 16, 32) is None
    L.kernel, L.cin, L.ntile_total = (3, 3, 3), 16, 1
    assert cp._wino_plan(L, 1, 4, 128, 160) == (9, 2, 1)                # 4 x 16 x 5 = 320 tiles
    assert cp._wino_plan(L, 1, 4, 32, 40) is None


def test_winograd_plan_only_names_instantiated_kernels():
    """Every (kernel depth, input channels, N tiles, map size) the heuristic can be asked about leads to a kernel instance
    the library has (cp.WINO_RING_INSTANCES / cp.WINO_INSTANCES mirror dispatch_wino's tables, which this test reads out
    of conv_wino.hip) -- e.g. a 16 -> 32 3x3x3 layer on a large map must not be sent to a two-N-tile ring kernel."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(cp.__file__), "csrc", "conv_wino.hip")).read()
    body = src[src.index("int dispatch_wino("):]
    ring = {tuple(int(x) for x in m) for m in re.findall(r"MV_R\((\d), (\d), (\d)\)", body)}
    plain = {(int(a), int(b)) for a, b in re.findall(r"MV_W\((\d), (\d), (?:true|false)\)", body)}
    assert ring == cp.WINO_RING_INSTANCES and plain == cp.WINO_INSTANCES

    class L:
        pass
    seen = 0
    for kd in (1, 3):
        for cin in (16, 32, 64):
            for ntile_total in (1, 2, 3, 4):
                for (B, D, H, W) in ((5, 1, 512, 640), (1, 8, 64, 80), (1, 4, 256, 320), (2, 4, 128, 160), (1, 1, 16, 32), (3, 1, 832, 1152)):
                    L.kernel, L.cin, L.ntile_total = (kd, 3, 3), cin, ntile_total
                    v = cp._wino_plan(L, B, D, H, W)
                    if v is None:
                        continue
                    seen += 1
                    word, mt, nt = v
                    fam, wpc = word & 0xff, word >> 8
                    assert mt == 2 and ntile_total % nt == 0
                    if fam == 8:
                        assert kd == 1 and (nt, cin // 16) in cp.WINO_INSTANCES, (kd, cin, ntile_total, v)
                    else:
                        mode = 0 if nt == 1 else (2 if wpc == 1 else 1)
                        assert fam == 9 and (mode, cin // 16, kd) in cp.WINO_RING_INSTANCES, (kd, cin, ntile_total, v)
    assert seen > 40


@pytest.mark.parametrize("cin,cout,k,s,p", [(16, 8, (1, 5, 5), (1, 2, 2), (0, 2, 2)), (32, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                            (16, 4, 3, 2, 1)])
def test_transposed_layer_is_the_input_gradient_of_a_strided_conv(cin, cout, k, s, p):
    """The transposed classes (3,2,1) and (5,2,2) with output_padding 1 are what the training path uses for the
    input gradient of the stride-2 convolutions (reg2d/reg3d 3x3, FPN 5x5): check against autograd."""
    torch.manual_seed(cin * 7 + cout)
    conv = torch.nn.Conv3d(cout, cin, k, stride=s, padding=p, bias=False)     # forward conv: cout -> cin channels
    x = torch.randn(2, cout, 2 if s in (1, 2) and k == 3 else 1, 8, 12, requires_grad=True)
    if k == 3:
        x = torch.randn(2, cout, 4, 8, 12, requires_grad=True)
    y = conv(x)
    gy = torch.randn_like(y)
    (gx,) = torch.autograd.grad(y, x, gy)
    layer = cp.ConvLayer(conv.weight, True, conv.stride, conv.padding)      # weight [cin_T = conv.out, cout_T = conv.in]
    got = run_layer(layer, cl(gy))
    assert got.shape == cl(gx).shape
    assert (got - cl(gx)).abs().max() <= 2e-5 * gx.abs().max()
    # repack after an in-place parameter update = a freshly built layer
    with torch.no_grad():
        conv.weight.mul_(0.5).add_(0.1)
    layer.repack(conv.weight)
    fresh = cp.ConvLayer(conv.weight, True, conv.stride, conv.padding)
    assert torch.equal(layer.wpk, fresh.wpk)


def test_tap_bias_gradient_of_the_fpn_gather():
    """train_ops.tap_bias_grad (the bias part of the re-associated FPN level's backward) against autograd through the
    PyTorch restatement of the gather-sum."""
    from mvster_amd.train_ops import tap_bias_grad
    from tests.conv_emulator import fpn_tail_gather_reference
    g = torch.Generator().manual_seed(11)
    NB, H, W, co = 2, 6, 10, 8
    G = torch.randn(NB, 1, H // 2, W // 2, 9 * co, generator=g, dtype=torch.float64)
    vb = torch.randn(9, co, generator=g, dtype=torch.float64, requires_grad=True)
    gP = torch.randn(NB, 1, H, W, co, generator=g, dtype=torch.float64)
    fpn_tail_gather_reference(G, vb, H, W).backward(gP)
    assert (tap_bias_grad(gP) - vb.grad).abs().max() <= 1e-10


def test_winograd_table_entry_is_dropped_for_a_layer_without_transformed_weights():
    """A tuning-table choice of the Winograd kernels (variant word 8 / 9) reaches a layer through its signature, which holds
    neither the padding nor the fused prob head: a layer of the same signature that is NOT Winograd-eligible (here: on the
    CPU no transformed weights exist at all, ``wpk_wino`` is None) keeps the heuristic choice instead of raising
    'conv_wino: layer not eligible' at call time."""
    torch.manual_seed(0)
    m = M.ConvBnReLU3D(16, 16, kernel_size=(1, 3, 3), stride=1, pad=(0, 1, 1))
    m.eval()
    L = cp._cbr3d(m)
    assert L.wpk_wino is None
    sig = cp.layer_signature(L, 1, 1, 96, 128, 0)
    saved = cp._TUNING, cp._FAMILIES
    try:
        cp._TUNING, cp._FAMILIES = {sig: [8, 2, 1]}, None
        assert cp.tuned_choice(L, 1, 1, 96, 128, 0) == ([8, 2, 1], "exact")
        variant = L._geom(1, 1, 96, 128, 0)[4]
        assert (variant & 0xff) not in (8, 9)
        # and the layer still computes (emulated direct kernel)
        x = torch.randn(1, 16, 1, 96, 128)
        with torch.no_grad():
            want = twin(O._CBR3d(16, 16, kernel_size=(1, 3, 3), stride=1, pad=(0, 1, 1)), m)(x)
        got = run_layer(L, cl(x))
        assert (got - cl(want)).abs().max() <= 2e-5 * want.abs().max()
    finally:
        cp._TUNING, cp._FAMILIES = saved


@pytest.mark.parametrize("cin,cout,kd", [(16, 16, 1), (32, 32, 3), (64, 64, 1), (64, 32, 3), (12, 16, 1)])
def test_bf16_split_weight_fragments(cin, cout, kd):
    """conv_plan.pack_b3 (operand of conv_b3.hip, variant 11): the three bf16 planes add up to the fp32 weight EXACTLY, and
    every value sits where the kernel's fragment rule reads it -- written out here element by element from the rule in the
    kernel's header (lane = (n, g), tap = 2 tp + (g >> 1), channel = 16 ch + 8 (g & 1) + e, tap 9 = zero)."""
    g = torch.Generator().manual_seed(cin + cout + kd)
    w = torch.randn(cout, cin, kd, 3, 3, generator=g) * torch.rand(cout, cin, kd, 3, 3, generator=g).exp()
    cin_pad = (cin + 15) // 16 * 16
    packed = cp.pack_b3(w, cin_pad, cout, kd)
    ntw = 1 if cout == 16 else 2
    nsplit = cout // (16 * ntw)
    assert packed.dtype == torch.bfloat16 and tuple(packed.shape) == (kd, cin_pad // 16, nsplit, 1024 * ntw * 8)
    frag = packed[..., :5 * ntw * 3 * 64 * 8].reshape(kd, cin_pad // 16, nsplit, 5, ntw, 3, 64, 8).double()
    assert (packed[..., 5 * ntw * 3 * 64 * 8:] == 0).all()
    total = frag.sum(5)                                       # w1 + w2 + w3 in fp64
    rng = np.random.RandomState(0)
    for _ in range(400):
        z, ch, ns, tp, j, lane, e = (rng.randint(n) for n in (kd, cin_pad // 16, nsplit, 5, ntw, 64, 8))
        n, gg = lane & 15, lane >> 4
        tap, ci, co = 2 * tp + (gg >> 1), 16 * ch + 8 * (gg & 1) + e, 16 * (ns * ntw + j) + n
        want = float(w[co, ci, z, tap // 3, tap % 3]) if (tap < 9 and ci < cin) else 0.0
        assert total[z, ch, ns, tp, j, lane, e].item() == want, (z, ch, ns, tp, j, lane, e)
    # the planes are ordered by magnitude: |w2| <= 2^-8 |w1|, |w3| <= 2^-16 |w1| (up to rounding at the boundaries)
    a1, a2, a3 = (frag[:, :, :, :, :, k].abs() for k in range(3))
    assert (a2 <= a1 * 2.0 ** -7.9 + 1e-40).all() and (a3 <= a1 * 2.0 ** -15.9 + 1e-40).all()
