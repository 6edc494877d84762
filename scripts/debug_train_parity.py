import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvster_amd import MVS4net, MVS4net_loss
from mvster_amd.synthetic import make_inputs, randomize_state
DEV = torch.device("cuda:0")
cfg = dict(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8, stage_splits=[8, 8, 4, 4],
           depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True, group_cor_dim=[8, 8, 4, 4], inverse_depth=True,
           mono=True, attn_temp=2, attn_fuse_d=True)
torch.manual_seed(1)
ref = MVS4net(**cfg); sd = randomize_state(ref.state_dict(), seed=4, prob_gain=4.0); ref.load_state_dict(sd)
nat = MVS4net(**cfg); nat.load_state_dict(sd)
ref.to(DEV).train(); nat.to(DEV).train(); ref.native_train = False
H, W, N, B = 128, 192, 3, 2
imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=8, batch=B)
imgs = [i.to(DEV) for i in imgs]; proj = {k: v.to(DEV) for k, v in proj.items()}; dv = dv.to(DEV)
g = torch.Generator().manual_seed(0)
gt, mask = {}, {}
for s in range(1, 5):
    hs, ws = H // 2 ** (4 - s), W // 2 ** (4 - s)
    gt["stage%d" % s] = (500 + 300 * torch.rand(B, hs, ws, generator=g)).to(DEV)
    mask["stage%d" % s] = (torch.rand(B, hs, ws, generator=g) > 0.2).float().to(DEV)
outs = []
for m in (ref, nat):
    out = m(imgs, proj, dv)
    loss, l1s, ots, _ = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10,
                        ot_eps=1, ot_continous=False, mono=True)
    loss.backward()
    outs.append(out)
    print("loss", loss.item(), [o.item() for o in ots])
for s in range(1, 5):
    a, b = outs[0]["stage%d" % s], outs[1]["stage%d" % s]
    print("stage", s, "attn max diff", (a["attn_weight"] - b["attn_weight"]).abs().max().item(),
          "depth differs at", (a["depth"] != b["depth"]).sum().item(), "of", a["depth"].numel(),
          "hypo max diff", (a["hypo_depth"] - b["hypo_depth"]).abs().max().item())
pr, pn = dict(ref.named_parameters()), dict(nat.named_parameters())
rows = []
for k in pr:
    if pr[k].grad is None: continue
    rows.append((((pr[k].grad - pn[k].grad).norm() / (pr[k].grad.norm() + 1e-30)).item(), pr[k].grad.norm().item(), k))
rows.sort(reverse=True)
for r in rows[:25]: print("%.3e  norm %.3e  %s" % r)
# conditioning of the step itself: the PyTorch-ROCm path again with the images perturbed by 1e-6 relative
ref.zero_grad(set_to_none=True)
g0 = {k: p.grad.clone() for k, p in nat.named_parameters() if p.grad is not None}
gn = torch.Generator().manual_seed(11)
imgs2 = [i * (1 + 1e-6 * torch.randn(i.shape, generator=gn).to(DEV)) for i in imgs]
gr = {k: v.grad for k, v in pr.items()}
ref2 = MVS4net(**cfg); ref2.load_state_dict(sd); ref2.to(DEV).train(); ref2.native_train = False
out = ref2(imgs2, proj, dv)
loss = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)[0]
loss.backward()
ref3 = MVS4net(**cfg); ref3.load_state_dict(sd); ref3.to(DEV).train(); ref3.native_train = False
out = ref3(imgs, proj, dv)
loss = MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)[0]
loss.backward()
p2, p3 = dict(ref2.named_parameters()), dict(ref3.named_parameters())
rows = []
for k in p3:
    if p3[k].grad is None: continue
    rows.append((((p3[k].grad - p2[k].grad).norm() / (p3[k].grad.norm() + 1e-30)).item(), p3[k].grad.norm().item(), k))
rows.sort(reverse=True)
print("--- PyTorch-ROCm path, images perturbed by 1e-6 relative:")
for r in rows[3:12]: print("%.3e  norm %.3e  %s" % r)
