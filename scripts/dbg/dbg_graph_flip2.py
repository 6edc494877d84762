import sys, os
sys.path.insert(0, '.')
import torch
from mvster_amd import MVS4net, MVS4net_loss, ops
from mvster_amd.graph import GraphedTrainStep
from mvster_amd.synthetic import make_inputs, randomize_state
DEV = torch.device("cuda:0")
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
def build():
    m = MVS4net(**cfg); m.load_state_dict(sd); m.to(DEV).train()
    return m, torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True)
if os.environ.get("TRACE"):
    orig = ops.warp_agg_bwd_cl
    def traced(*a, **k):
        torch.cuda.synchronize(); print("  warp bwd in", tuple(a[0].shape), flush=True)
        r = orig(*a, **k)
        torch.cuda.synchronize(); print("  warp bwd ok", flush=True)
        return r
    ops.warp_agg_bwd_cl = traced
m1, o1 = build()
for _ in range(6):
    o1.zero_grad(set_to_none=False)
    loss = loss_fn(m1(imgs, proj, dv), gt, mask)[0]
    loss.backward(); o1.step()
print("eager done", loss.item(), flush=True)
m2, o2 = build()
step = GraphedTrainStep(m2, o2, loss_fn, imgs, proj, dv, gt, mask, warmup=3, capture=os.environ.get("CAPTURE", "1") == "1")
for it in range(4):
    print("step", it, step().item(), flush=True)
print("flip", step(imgs=[i.flip(-1) for i in imgs]).item(), flush=True)
