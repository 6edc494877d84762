import sys, os
sys.path.insert(0, '.')
import torch
from mvster_amd import MVS4net, MVS4net_loss, ops
from mvster_amd.graph import GraphedTrainStep
from mvster_amd.synthetic import make_inputs, randomize_state
DEV = torch.device("cuda:0")
ops.SORTED_SCATTER = os.environ.get("SORTED", "1") == "1"
cfg = dict(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8, stage_splits=[8, 8, 4, 4],
           depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True, group_cor_dim=[8, 8, 4, 4], inverse_depth=True,
           mono=True, attn_temp=2, attn_fuse_d=True)
torch.manual_seed(4)
sd = randomize_state(MVS4net(**cfg).state_dict(), seed=6, prob_gain=4.0)
H, W, N, B = 128, 192, 3, 2
imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=3, batch=B)
imgs = [i.to(DEV) for i in imgs]
proj = {k: v.to(DEV) for k, v in proj.items()}
dv = dv.to(DEV)
g = torch.Generator().manual_seed(0)
gt = {"stage%d" % s: (500 + 300 * torch.rand(B, H // 2 ** (4 - s), W // 2 ** (4 - s), generator=g)).to(DEV) for s in range(1, 5)}
mask = {k: (torch.rand(v.shape, generator=g) > 0.2).float().to(DEV) for k, v in gt.items()}
def loss_fn(o, g_, m_):
    return MVS4net_loss(o, g_, m_, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)
m = MVS4net(**cfg); m.load_state_dict(sd); m.to(DEV).train()
o = torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True)
if os.environ.get("EAGER"):
    for it in range(8):
        o.zero_grad(set_to_none=True)
        im = imgs if it < 5 else [i.flip(-1) for i in imgs]
        loss = loss_fn(m(im, proj, dv), gt, mask)[0]
        loss.backward(); o.step()
        print("eager", it, loss.item(), flush=True)
    sys.exit(0)
step = GraphedTrainStep(m, o, loss_fn, imgs, proj, dv, gt, mask, warmup=3)
for it in range(4):
    print("replay", it, step().item(), flush=True)
print("flip", step(imgs=[i.flip(-1) for i in imgs]).item(), flush=True)
print("again", step().item(), flush=True)
