"""mvster_amd.loss (PyTorch ops, device-agnostic) against the reference's golden vectors."""
import torch

from mvster_amd.loss import Blend_loss, MVS4net_loss, sinkhorn


def test_sinkhorn_golden(golden):
    g = golden("g8_sinkhorn")
    T, loss = sinkhorn(g.t("gt"), g.t("hypo"), g.t("attn"), g.t("mask"), iters=10, eps=1)
    assert (T - g.t("T")).abs().max() <= 1e-6
    assert abs(loss.item() - float(g.np("loss"))) <= 1e-6


def test_loss_on_golden_train_outputs(golden):
    g = golden("g6_train")
    inputs, gt, mask = {}, {}, {}
    for s in range(1, 5):
        st = {"depth": g.t("stage%d_depth" % s), "hypo_depth": g.t("stage%d_hypo_depth" % s),
              "attn_weight": g.t("stage%d_attn_weight" % s)}
        if s > 1:
            st["mono_depth"] = g.t("stage%d_mono_depth" % s)
        inputs["stage%d" % s] = st
        gt["stage%d" % s] = g.t("depth_gt_stage%d" % s)
        mask["stage%d" % s] = g.t("mask_stage%d" % s)
    loss, l1s, ots, rng = MVS4net_loss(inputs, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True,
                                       ot_iter=10, ot_eps=1, ot_continous=False, mono=True)
    assert abs(loss.item() - float(g.np("loss"))) <= 1e-5 * abs(float(g.np("loss")))
    assert torch.allclose(torch.stack(ots), g.t("ot"), rtol=1e-5)
    assert torch.allclose(torch.stack(l1s), g.t("l1"), rtol=1e-5)
    out = Blend_loss(inputs, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10,
                     depth_max=torch.tensor([935.0, 935.0]), depth_min=torch.tensor([425.0, 425.0]), mono=True)
    assert len(out) == 7 and abs(out[0].item() - loss.item()) <= 1e-6
