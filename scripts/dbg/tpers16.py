import sys
sys.path.insert(0, '.')
import torch
from mvster_amd.conv_plan import ConvLayer, SKIP_ADD
dev = torch.device("cuda:0")
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
w = torch.randn(16, 8, 1, 3, 3, generator=g).to(dev) * 0.1
for shape in [(2, 4, 256, 320), (2, 4, 128, 160), (2, 8, 64, 80), (2, 8, 32, 40)]:
    B, D, H, W = shape
    x = torch.randn(B, D, H, W, 16, generator=g).to(dev)
    skip = torch.randn(B, D, 2 * H, 2 * W, 8, generator=g).to(dev)
    L = ConvLayer(w, True, (1, 2, 2), (0, 1, 1), cin_pad=16)
    for sk in (None, skip):
        kw = {} if sk is None else dict(skip=sk, skip_mode=SKIP_ADD)
        a = L(x, **kw)
        b = L(x, tiles=(2, 1, 5), **kw)
        c = L(x, tiles=(1, 1, 0), **kw)
        err = (a - b).abs().max().item() / a.abs().max().item()
        errc = (a - c).abs().max().item() / a.abs().max().item()
        ta = timeit(lambda: L(x, **kw)); tb = timeit(lambda: L(x, tiles=(2, 1, 5), **kw))
        print(shape, "skip" if sk is not None else "    ", "default %.1f us  tpers %.1f us  rel diff %.1e (direct vs default %.1e)" % (ta, tb, err, errc), flush=True)
