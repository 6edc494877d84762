"""Every weight-gradient call of one eager config-4 training step (512x640, 5 views, B = 2): kernel, time between HIP events,
TFLOP/s on direct-form FLOPs; slowest first.  GPU only."""
import sys
sys.path.insert(0, '.')
import torch
from mvster_amd import ops, _lib
from bench import SHIPPED, load_weights
from mvster_amd import MVS4net, MVS4net_loss
from mvster_amd.synthetic import make_inputs
dev = torch.device("cuda:0")
rows = []
orig = ops.conv_wgrad
def logged(x_cl, gy_cl, kernel, stride, padding, **kw):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(x_cl, gy_cl, kernel, stride, padding, **kw)
    e1.record(); torch.cuda.synchronize()
    rows.append((e0.elapsed_time(e1) * 1e3, tuple(x_cl.shape), tuple(gy_cl.shape), kernel, stride, _lib.last_kernel(), kw.get("mirrored", False)))
    return r
ops.conv_wgrad = logged
model = MVS4net(**SHIPPED); model.load_state_dict(load_weights(), strict=True); model.to(dev).train()
imgs, proj, dv = make_inputs(5, 512, 640, seed=0, device=dev, batch=2)
g = torch.Generator().manual_seed(0)
gt, mask = {}, {}
for s in range(1, 5):
    hs, ws = 512 // 2 ** (4 - s), 640 // 2 ** (4 - s)
    gt["stage%d" % s] = (500 + 300 * torch.rand(2, hs, ws, generator=g)).to(dev)
    mask["stage%d" % s] = (torch.rand(2, hs, ws, generator=g) > 0.2).float().to(dev)
for it in range(2):
    rows.clear()
    model.zero_grad(set_to_none=True)
    loss = MVS4net_loss(model(imgs, proj, dv), gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)[0]
    loss.backward()
torch.cuda.synchronize()
for r in sorted(rows, key=lambda r: -r[0])[:26]:
    flops = 2.0 * r[1][-1] * r[2][-1] * r[3][0] * r[3][1] * r[3][2] * (r[2][0] * r[2][1] * r[2][2] * r[2][3])
    print("%7.1f us %5.1f TF/s  x %-24s gy %-24s k %s s %s %s %s" % (r[0], flops / r[0] / 1e6, r[1], r[2], r[3], r[4], r[5], "mirrored" if r[6] else ""))
print("total %.1f us over %d calls" % (sum(r[0] for r in rows), len(rows)))
