import sys
sys.path.insert(0, '.')
import torch
from mvster_amd import conv_plan
from bench import SHIPPED, load_weights
from mvster_amd import MVS4net, MVS4net_loss
from mvster_amd.synthetic import make_inputs
seen = {}
orig = conv_plan.tuned_choice
def logged(layer, B, Di, Hi, Wi, skip_mode):
    r = orig(layer, B, Di, Hi, Wi, skip_mode)
    seen[conv_plan.layer_signature(layer, B, Di, Hi, Wi, skip_mode)] = r
    return r
conv_plan.tuned_choice = logged
dev = torch.device("cuda:0")
model = MVS4net(**SHIPPED); model.load_state_dict(load_weights(), strict=True); model.to(dev).train()
imgs, proj, dv = make_inputs(5, 512, 640, seed=0, device=dev, batch=2)
g = torch.Generator().manual_seed(0)
gt, mask = {}, {}
for s in range(1, 5):
    hs, ws = 512 // 2 ** (4 - s), 640 // 2 ** (4 - s)
    gt["stage%d" % s] = (500 + 300 * torch.rand(2, hs, ws, generator=g)).to(dev)
    mask["stage%d" % s] = (torch.rand(2, hs, ws, generator=g) > 0.2).float().to(dev)
loss = MVS4net_loss(model(imgs, proj, dv), gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)[0]
loss.backward()
torch.cuda.synchronize()
for sig, (val, kind) in sorted(seen.items()):
    print("%-8s %-52s %s" % (kind, sig, val))
