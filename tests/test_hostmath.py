"""The per-element arithmetic shared with the HIP kernels (mvster_amd/csrc/mvster_math.h),
compiled for the host and checked against the golden vectors.  This is what lets the
arithmetic of the GPU kernels be verified in the CPU-only container; the kernels' indexing
and launch geometry are checked by the -m gpu tests."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import mvs4_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm():
    subprocess.check_call([os.path.join(HERE, "hostmath", "build.sh")])
    return ctypes.CDLL(os.path.join(HERE, "hostmath", "libhostmath.so"))


def fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ref_rt(src, ref):
    P = torch.matmul(torch.from_numpy(src), torch.inverse(torch.from_numpy(ref)))[0]
    return c(np.concatenate([P[:3, :3].reshape(-1).numpy(), P[:3, 3].numpy()]))


@pytest.mark.parametrize("case,src,ref", [("a", "a_src", "a_ref"), ("b", "a_src", "a_ref"), ("c", "c_src", "a_ref")])
def test_warp_given_projection(hm, golden, case, src, ref):
    """With the reference's own relative projection, the per-pixel math agrees to fp32 rounding."""
    g = golden("g1_warp")
    fea = c(g.np(case + "_fea")[0])
    depth = c(g.np("a_depth")[0])
    C, Hs, Ws = fea.shape
    D, Hr, Wr = depth.shape
    out = np.zeros((C, D, Hr, Wr), np.float32)
    hm.hm_warp(fp(fea), fp(ref_rt(g.np(src), g.np(ref))), fp(depth), fp(out), C, D, Hr, Wr, Hs, Ws)
    want = g.np(case + "_out")[0]
    assert np.abs(out - want).max() <= 5e-7 * np.abs(want).max()


def test_warp_z_zero(hm, golden):
    g = golden("g1_warp")
    fea = c(g.np("c_fea")[0])
    depth = c(g.np("d_depth")[0])
    rt = c(np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]))
    C, Hs, Ws = fea.shape
    D, Hr, Wr = depth.shape
    out = np.zeros((C, D, Hr, Wr), np.float32)
    hm.hm_warp(fp(fea), fp(rt), fp(depth), fp(out), C, D, Hr, Wr, Hs, Ws)
    assert np.abs(out - g.np("d_out")[0]).max() <= 5e-7 * np.abs(fea).max()


def test_relative_projection_close_to_fp32_lapack(hm):
    """fp64 Gauss-Jordan vs the reference's fp32 torch.inverse: equal up to the fp32 path's own error."""
    from mvster_amd.synthetic import make_inputs
    _, proj, _ = make_inputs(5, 512, 640, seed=0)
    for stage in ("stage1", "stage4"):
        pm = proj[stage][0].numpy()
        for v in range(1, 5):
            rt = np.zeros(12, np.float32)
            assert hm.hm_relative_projection(fp(c(pm[0])), fp(c(pm[v])), fp(rt)) == 0
            Pr = O.compose_projection(proj[stage][:, 0]).double()
            Ps = O.compose_projection(proj[stage][:, v]).double()
            P = torch.matmul(Ps, torch.inverse(Pr))[0]
            want = np.concatenate([P[:3, :3].reshape(-1).numpy(), P[:3, 3].numpy()])
            # against the float64 evaluation of the same formula (K @ E composed in fp32 by both;
            # a 1-ulp difference there is amplified a few times by the cancellation in rot[:, 2])
            assert np.all(np.abs(rt - want) <= 5e-7 * np.maximum(np.abs(want), 1.0))


def test_singular_reference_projection_is_loud(hm):
    """torch.inverse raises on a singular ref_proj (mvs4net_utils.py:25); the kernel arithmetic marks the whole
    relative projection NaN instead of returning a silently wrong warp."""
    from mvster_amd.synthetic import make_inputs
    _, proj, _ = make_inputs(2, 64, 64, seed=0)
    pm = proj["stage1"][0].numpy().copy()
    pm[0, 0, 2, :] = 0.0                       # third extrinsic row zero: K @ E has rank 2
    rt = np.zeros(12, np.float32)
    assert hm.hm_relative_projection(fp(c(pm[0])), fp(c(pm[1])), fp(rt)) == -1
    assert np.isnan(rt).all()


def test_schedulers(hm, golden):
    g = golden("g5_sched")
    dv = g.np("dv")
    for b in range(2):
        out = np.zeros((8, 60), np.float32)
        hm.hm_init_range(ctypes.c_float(dv[b, 0]), ctypes.c_float(dv[b, 1]), fp(out), 8, 60, 1)
        assert np.array_equal(out.reshape(8, 6, 10), g.np("init_inverse_8")[b])
        hm.hm_init_range(ctypes.c_float(dv[b, 0]), ctypes.c_float(dv[b, 1]), fp(out), 8, 60, 0)
        assert np.array_equal(out.reshape(8, 6, 10), g.np("init_range_8")[b])
        for D in (8, 4):
            out = np.zeros((D, 24, 40), np.float32)
            hm.hm_schedule_inverse(fp(c(g.np("inv_min")[b])), fp(c(g.np("inv_max")[b])), fp(out), D, 24, 40)
            want = g.np("sched_inverse_%d" % D)[b]
            assert np.abs(out - want).max() <= 5e-7 * np.abs(want).max()  # <= 3 ulp: ATen contracts its lerp differently
        out = np.zeros((4, 24, 40), np.float32)
        hm.hm_schedule_linear(fp(c(g.np("cur_depth")[b])), ctypes.c_float(g.np("itv")[b]), fp(out), 4, 24, 40)
        want = g.np("sched_range_4")[b]
        assert np.abs(out - want).max() <= 5e-7 * np.abs(want).max()  # <= 3 ulp: ATen contracts its lerp differently


@pytest.mark.parametrize("name", ["d8_s0", "d4_s2", "d4_s3_ties", "d8_s1_b2"])
def test_select(hm, golden, name):
    g = golden("g4_select")
    logits = g.np(name + "_logits")
    hypo = g.np(name + "_hypo")
    B, D, h, w = logits.shape
    s = int(g.np(name + "_stage_idx"))
    up = 2 ** (3 - s)
    for b in range(B):
        attn = np.zeros((D, h, w), np.float32)
        depth = np.zeros((h, w), np.float32)
        conf = np.zeros((h, w), np.float32)
        imin = np.zeros((h, w), np.float32)
        imax = np.zeros((h, w), np.float32)
        hm.hm_select(fp(c(logits[b])), None, None, None, 0, fp(c(hypo[b])), fp(attn), fp(depth), fp(conf), fp(imin),
                     fp(imax), None, D, h * w, ctypes.c_float(0.5))
        assert np.abs(attn - g.np(name + "_attn_weight")[b]).max() <= 3e-7
        # ties included: the first maximum is selected, like torch.max
        want_d = g.np(name + "_depth")[b]
        top2 = np.sort(g.np(name + "_attn_weight")[b], axis=0)[-2:]
        clear = (top2[1] - top2[0] > 1e-6) | (top2[1] == top2[0])
        assert np.array_equal(depth[clear], want_d[clear])
        assert np.abs(imin - g.np(name + "_inverse_min_depth")[b])[clear].max() <= 1e-9
        assert np.abs(imax - g.np(name + "_inverse_max_depth")[b])[clear].max() <= 1e-9
        big = np.zeros((h * up, w * up), np.float32)
        hm.hm_upsample(fp(conf), fp(big), h, w, h * up, w * up)
        assert np.abs(big - g.np(name + "_photometric_confidence")[b]).max() <= 5e-7


def test_select_with_prob_head(hm):
    rng = np.random.RandomState(0)
    D, hw, CF = 4, 50, 8
    feat = c(rng.randn(D, hw, CF))
    w = c(rng.randn(CF))
    bias = c(np.array([0.3]))
    hypo = c(500 + 100 * rng.rand(D, hw))
    attn = np.zeros((D, hw), np.float32)
    depth = np.zeros(hw, np.float32)
    lo = np.zeros((D, hw), np.float32)
    hm.hm_select(None, fp(feat), fp(w), fp(bias), CF, fp(hypo), fp(attn), fp(depth), None, None, None, fp(lo), D, hw,
                 ctypes.c_float(0.5))
    want = feat @ w + 0.3
    assert np.abs(lo - want).max() <= 1e-5
    assert np.abs(attn - torch.softmax(torch.from_numpy(want), 0).numpy()).max() <= 1e-6


def _run_select(fn, logits, feat, w, bias, hypo, inverse):
    D, hw = hypo.shape
    attn = np.zeros((D, hw), np.float32)
    depth, conf, imin, imax = (np.zeros(hw, np.float32) for _ in range(4))
    lo = np.zeros((D, hw), np.float32)
    fn(None if logits is None else fp(logits), None if feat is None else fp(feat), None if w is None else fp(w),
       None if bias is None else fp(bias), 0 if feat is None else feat.shape[-1], fp(hypo), fp(attn), fp(depth), fp(conf),
       fp(imin) if inverse else None, fp(imax) if inverse else None, fp(lo) if feat is not None else None, D, hw,
       ctypes.c_float(0.5))
    return attn, depth, conf, imin, imax, lo


@pytest.mark.parametrize("D,with_head", [(3, False), (8, False), (16, True), (4, True)])
def test_select_any_equals_the_register_form_bit_for_bit(hm, D, with_head):
    """mv::select_pixel_any (any number of hypotheses, through memory) against mv::select_pixel (D <= 16, registers) where
    both apply: every output identical, exact ties included."""
    rng = np.random.RandomState(D)
    hw, CF = 97, 8
    hypo = c(np.sort(400 + 300 * rng.rand(D, hw), axis=0))
    logits = c(rng.randn(D, hw))
    logits[1, :9] = logits[0, :9] = logits.max(0)[:9] + 1       # exact ties between the first two hypotheses
    feat = c(rng.randn(D, hw, CF)) if with_head else None
    w = c(rng.randn(CF)) if with_head else None
    bias = c(np.array([0.1])) if with_head else None
    a = _run_select(hm.hm_select, None if with_head else logits, feat, w, bias, hypo, True)
    b = _run_select(hm.hm_select_any, None if with_head else logits, feat, w, bias, hypo, True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("D", [24, 48, 192])
def test_select_any_many_hypotheses_vs_torch(hm, D):
    """More hypotheses than the register form holds (a free --ndepths): softmax / first-max argmax / gather / inverse bounds
    of mvs4net_utils.py:1068-1088 restated with torch."""
    rng = np.random.RandomState(D)
    hw = 61
    hypo = c(np.sort(400 + 300 * rng.rand(D, hw), axis=0))
    logits = c(2 * rng.randn(D, hw))
    logits[5, :7] = logits[2, :7] = logits.max(0)[:7] + 1
    attn, depth, conf, imin, imax, _ = _run_select(hm.hm_select_any, logits, None, None, None, hypo, True)
    want = torch.softmax(torch.from_numpy(logits), 0)
    assert np.abs(attn - want.numpy()).max() <= 3e-7
    idx = torch.from_numpy(attn).max(0)[1]                      # ATen max: the first maximum of the probabilities we produced
    assert np.array_equal(depth, np.take_along_axis(hypo, idx.numpy()[None], 0)[0])
    assert np.array_equal(idx.numpy()[:7], np.full(7, 2))
    assert np.array_equal(conf, attn.max(0))
    itv = 1.0 / hypo[2] - 1.0 / hypo[1]
    assert np.abs(imin - (1.0 / depth + 0.5 * itv)).max() <= 1e-9
    assert np.abs(imax - (1.0 / depth - 0.5 * itv)).max() <= 1e-9


def test_multiply_shift_division(hm):
    """FastDiv (mvster_math.h), the division by launch constants used in the kernels' index arithmetic: exact for every
    dividend below 2^31 -- checked here for the divisors the path produces (tile counts, patch widths, image sizes) and
    awkward ones, over dense low ranges, strided sweeps of the whole range and the top end."""
    hm.hm_fastdiv_mismatches.restype = ctypes.c_long
    hm.hm_fastdiv_mismatches.argtypes = [ctypes.c_uint] * 4
    hm.hm_fastdiv.restype = ctypes.c_uint
    hm.hm_fastdiv.argtypes = [ctypes.c_uint] * 2
    top = 2 ** 31
    for d in (1, 2, 3, 5, 6, 7, 9, 10, 18, 19, 20, 34, 35, 36, 40, 64, 80, 100, 131, 160, 200, 240, 320, 400, 480, 640, 800,
              960, 1600, 1920, 4096, 65535, 65537, 1000003, 2 ** 30 - 1, 2 ** 30 + 1, 2 ** 31 - 1):
        assert hm.hm_fastdiv_mismatches(d, 0, 200000, 1) == 0, d
        assert hm.hm_fastdiv_mismatches(d, 0, top, 104729) == 0, d
        assert hm.hm_fastdiv_mismatches(d, top - 200000, top, 1) == 0, d
    assert hm.hm_fastdiv(2 ** 31 - 1, 7) == (2 ** 31 - 1) // 7
