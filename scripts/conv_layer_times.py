"""Every convolution call (forward and input gradient) of one eager config-4 training step (512x640, 5 views, B = 2): layer
signature, where the plan's choice came from, kernel, time between HIP events, TFLOP/s on direct-form FLOPs; slowest first.
GPU only."""
import sys
sys.path.insert(0, '.')
import torch
from mvster_amd import _lib
import mvster_amd.conv_plan as cp
from bench import SHIPPED, load_weights
from mvster_amd import MVS4net, MVS4net_loss
from mvster_amd.synthetic import make_inputs
dev = torch.device("cuda:0")
rows = []
orig = cp.ConvLayer.__call__
def logged(self, x, skip=None, skip_mode=cp.SKIP_NONE, tiles=None):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(self, x, skip=skip, skip_mode=skip_mode, tiles=tiles)
    e1.record(); torch.cuda.synchronize()
    B, Di, Hi, Wi = x.shape[:4]
    sig = cp.layer_signature(self, B, Di, Hi, Wi, skip_mode)
    how = cp.tuned_choice(self, B, Di, Hi, Wi, skip_mode)[1] or "heuristic"
    taps = self.kernel[0] * self.kernel[1] * self.kernel[2] if hasattr(self, "kernel") else 0
    vox = y.numel() // y.shape[-1] if not getattr(self, "transposed", False) else x.numel() // x.shape[-1]
    flops = 2.0 * vox * taps * self.cin * self.cout
    rows.append((e0.elapsed_time(e1) * 1e3, sig, how, _lib.last_kernel(), flops))
    return y
cp.ConvLayer.__call__ = logged
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
for r in sorted(rows, key=lambda r: -r[0])[:int(sys.argv[1]) if len(sys.argv) > 1 else 40]:
    print("%7.1f us %6.1f TF/s  %-52s %-10s %s" % (r[0], r[4] / r[0] / 1e6, r[1], r[2], r[3]))
print("total %.1f us over %d calls" % (sum(r[0] for r in rows), len(rows)))
