import torch, sys
sys.path.insert(0, '.')
from mvster_amd.optim import FusedAdam
DEV='cuda:0'
g = torch.Generator().manual_seed(1)
shapes = [(8, 3, 3, 3), (8,), (64, 64, 3, 3), (1,), (5, 7), (1025,), (4096, 3)] * 43
for wd in (0.0, 0.01):
    base = [torch.randn(s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    pb = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    pc = [torch.nn.Parameter(b.clone().to(DEV)) for b in base]
    oa = FusedAdam(pa, lr=1e-2, weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=1e-2, weight_decay=wd)
    oc = torch.optim.Adam(pc, lr=1e-2, weight_decay=wd, fused=True)
    for it in range(5):
        grads = [torch.randn(s, generator=g).to(DEV) * (10.0 ** (it - 2)) for s in shapes]
        for o, ps in ((oa, pa), (ob, pb), (oc, pc)):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone()
            if it == 3:
                o.param_groups[0]["lr"] = 3e-3
            o.step()
        e = [((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item() for a, b in zip(pa, pb)]
        e2 = [((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item() for a, b in zip(pc, pb)]
        k = max(range(len(e)), key=lambda i: e[i])
        print(wd, it, "mine vs torch", max(e), "idx", k, shapes[k], " torch fused vs torch", max(e2))
