"""mvster_amd.loss host logic (masks, range ratios, stage weighting, Blend_loss's error figures) against the
reference's golden vectors.  The product's OT term is the fused HIP kernel and refuses CPU tensors; here it is replaced
by the oracle's tensor-level ``sinkhorn`` (pinned by G8 / G8b in test_oracle_golden.py) so that the surrounding logic
can be checked without a GPU."""
import pytest
import torch

from mvster_amd import loss as L
from mvster_amd.loss import Blend_loss, MVS4net_loss
from oracle.mvs4_oracle import sinkhorn
from tests.test_oracle_golden import G9_CASES, _g6_train_stage_dicts


@pytest.fixture
def tensor_level_ot(monkeypatch):
    monkeypatch.setattr(L, "sinkhorn_loss", lambda gt, hypo, attn, mask, iters, eps=1, continuous=False:
                        sinkhorn(gt, hypo, attn, mask, iters, eps, continuous)[1])


def test_ot_term_has_no_cpu_fallback(golden):
    g = golden("g8_sinkhorn")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.sinkhorn_loss(g.t("gt"), g.t("hypo"), g.t("attn"), g.t("mask"), 10)


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
