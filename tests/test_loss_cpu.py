"""mvster_amd.loss host logic (stage weighting, the terms' bookkeeping, Blend_loss's error figures) against the
reference's golden vectors.  The product's per-stage terms are fused HIP kernels and refuse CPU tensors; here they are
replaced by a tensor-level restatement of models/MVS4Net.py:131-151 built on the oracle's ``sinkhorn`` (pinned by
G8 / G8b in test_oracle_golden.py) so that the surrounding logic can be checked without a GPU.  The kernels themselves
meet the same golden values in tests/test_gpu_train.py::test_losses_vs_reference_values_on_device."""
import pytest
import torch

from mvster_amd import loss as L
from mvster_amd.loss import Blend_loss, MVS4net_loss
from oracle.mvs4_oracle import sinkhorn
from tests.test_oracle_golden import G9_CASES, _g6_train_stage_dicts


def oracle_stage_losses(gt, hypo, attn, mask, mono_depth=None, iters=3, eps=1, continuous=False, inverse=False):
    """(l1, ot, out_of_range_ratio) of one stage as tensor expressions (models/MVS4Net.py:131-151)."""
    m = mask > 0.5
    if mono_depth is not None:
        l1 = torch.nn.functional.l1_loss(mono_depth[m], gt[m], reduction="mean")
    else:
        l1 = torch.zeros((), dtype=torch.float32, device=gt.device)
    t = (lambda x: 1 / x) if inverse else (lambda x: x)
    itv = (t(hypo[:, 2]) - t(hypo[:, 1])).abs()
    oor = ((t(hypo) - t(gt).unsqueeze(1)).abs() <= itv.unsqueeze(1)).sum(1) == 0
    return l1, sinkhorn(gt, hypo, attn, m, iters, eps, continuous)[1], oor[m].float().mean()


def oracle_stage_losses_total(gt, hypo, attn, mask, mono_depth, total, iters, eps, continuous, inverse, w_l1, w_ot, w_stage):
    """The product's fused per-stage call (terms + the weighted running total, models/MVS4Net.py:151) as tensor expressions."""
    l1, ot, ratio = oracle_stage_losses(gt, hypo, attn, mask, mono_depth, iters, eps, continuous, inverse)
    total = torch.zeros((), dtype=torch.float32, device=gt.device) if total is None else total
    return l1, ot, ratio, total + w_stage * (w_l1 * l1 + w_ot * ot)


@pytest.fixture
def tensor_level_ot(monkeypatch):
    monkeypatch.setattr(L, "stage_losses", oracle_stage_losses)
    monkeypatch.setattr(L, "stage_losses_total", oracle_stage_losses_total)


def test_ot_term_has_no_cpu_fallback(golden):
    g = golden("g8_sinkhorn")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.sinkhorn_loss(g.t("gt"), g.t("hypo"), g.t("attn"), g.t("mask"), 10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.stage_losses(g.t("gt"), g.t("hypo"), g.t("attn"), g.t("mask"), iters=10)
    inputs, gt, mask = _g6_train_stage_dicts(golden)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        MVS4net_loss(inputs, gt, mask, inverse_depth=True, ot_iter=10, mono=True)


@pytest.mark.parametrize("name", sorted(G9_CASES))
def test_losses_vs_reference_values(golden, tensor_level_ot, name):
    """MVS4net_loss and Blend_loss (MVS4Net.py:113-206) on the reference's own stage outputs: totals, per-stage terms,
    out-of-range ratios, normalised end-point error and inlier percentages."""
    g = golden("g9_losses")
    inputs, gt, mask = _g6_train_stage_dicts(golden)
    kw = G9_CASES[name]
    total, l1s, ots, rng = MVS4net_loss(inputs, gt, mask, **kw)
    assert abs(total.item() - float(g.np("mvs4_%s_total" % name))) <= 1e-5 * abs(float(g.np("mvs4_%s_total" % name)))
    assert torch.allclose(torch.stack(l1s), g.t("mvs4_%s_l1" % name), rtol=1e-5)
    assert torch.allclose(torch.stack(ots), g.t("mvs4_%s_ot" % name), rtol=1e-5)
    assert torch.allclose(torch.stack(rng), g.t("mvs4_%s_range" % name), rtol=1e-6)
    r = Blend_loss(inputs, gt, mask, depth_max=g.t("depth_max"), depth_min=g.t("depth_min"), **kw)
    assert len(r) == 7
    assert abs(r[0].item() - float(g.np("blend_%s_total" % name))) <= 1e-5 * abs(float(g.np("blend_%s_total" % name)))
    assert torch.allclose(torch.stack(r[1]), g.t("blend_%s_l1" % name), rtol=1e-5)
    assert torch.allclose(torch.stack(r[2]), g.t("blend_%s_ot" % name), rtol=1e-5)
    assert torch.allclose(torch.stack(r[3]), g.t("blend_%s_range" % name), rtol=1e-6)
    assert torch.allclose(torch.stack(list(r[4:])), g.t("blend_%s_epe_err3_err1" % name), rtol=1e-5)


def test_loss_on_golden_train_outputs(golden, tensor_level_ot):
    g = golden("g6_train")
    inputs, gt, mask = _g6_train_stage_dicts(golden)
    loss, l1s, ots, rng = MVS4net_loss(inputs, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True,
                                       ot_iter=10, ot_eps=1, ot_continous=False, mono=True)
    assert abs(loss.item() - float(g.np("loss"))) <= 1e-5 * abs(float(g.np("loss")))
    assert torch.allclose(torch.stack(ots), g.t("ot"), rtol=1e-5)
    assert torch.allclose(torch.stack(l1s), g.t("l1"), rtol=1e-5)
