"""Weight-gradient time of the staged-kernel layers against the number of partial-sum slots (= workgroups) the caller gives
them (ops.WGRAD_MAX_SLOTS): where one resident round of workgroups beats a larger grid.  GPU only."""
import sys
sys.path.insert(0, '.')
import torch
from mvster_amd import ops, _lib
dev = torch.device("cuda:0")
layers = [((10, 1, 512, 640, 8), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          ((10, 1, 512, 640, 4), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          ((2, 4, 512, 640, 4), 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
          ((10, 1, 512, 640, 8), 16, (1, 5, 5), (1, 2, 2), (0, 2, 2)),
          ((10, 1, 256, 320, 16), 32, (1, 5, 5), (1, 2, 2), (0, 2, 2)),
          ((10, 1, 128, 160, 32), 64, (1, 5, 5), (1, 2, 2), (0, 2, 2)),
          ((10, 1, 256, 320, 64), 80, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
          ((2, 4, 128, 160, 32), 64, (1, 3, 3), (1, 2, 2), (0, 1, 1))]
for xs, co, k, s, p in layers:
    x = torch.randn(xs, device=dev)
    os_ = [xs[0]] + [(xs[1 + i] + 2 * p[i] - k[i]) // s[i] + 1 for i in range(3)] + [co]
    gy = torch.randn(os_, device=dev)
    line = "x %-24s co %2d k %s s %s :" % (xs, co, k, s)
    for slots in (32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 2048):
        ops.WGRAD_MAX_SLOTS = slots
        for _ in range(3):
            ops.conv_wgrad(x, gy, k, s, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_wgrad(x, gy, k, s, p)
        e1.record(); torch.cuda.synchronize()
        line += " %d:%.1f" % (slots, e0.elapsed_time(e1) * 50)
    print(line, _lib.last_kernel())
