#!/usr/bin/env python3
"""BatchNorm kernels in isolation at the training step's shapes: achieved GB/s of the four passes (stats, apply, backward
reduce, backward apply) -- algorithmic bytes = reads + writes of the tensor(s)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from mvster_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


shapes = [((2, 4, 512, 640, 8), 1), ((2, 4, 256, 320, 16), 1), ((2, 4, 128, 160, 32), 1), ((2, 4, 64, 80, 64), 1),
          ((10, 1, 512, 640, 8), 5), ((10, 1, 256, 320, 16), 5), ((10, 1, 64, 80, 64), 5),
          ((2, 8, 64, 80, 8), 1), ((2, 8, 8, 10, 64), 1)]
for shape, groups in shapes:
    C = shape[-1]
    x = torch.randn(shape, device=dev)
    gy = torch.randn(shape, device=dev)
    w, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    pack = ops.bn_batch_stats(x, w, b, rm, rv, 1e-5, 0.1, groups, num_batches_tracked=nbt)
    mb = x.numel() * 4 / 1e6
    t_stats = timeit(lambda: ops.bn_batch_stats(x, w, b, rm, rv, 1e-5, 0.1, groups, num_batches_tracked=nbt))
    t_fwd = timeit(lambda: ops.bn_relu_fwd(x, pack[3], pack[4], True, groups))
    t_bwd = timeit(lambda: ops.bn_relu_bwd(x, gy, pack[3], pack[4], pack[0], pack[2], True, groups))
    print("%-24s g=%d %7.1f MB  stats %6.1f us %5.0f GB/s   apply %6.1f us %5.0f GB/s   bwd (reduce+apply) %6.1f us %5.0f GB/s"
          % (shape, groups, mb, t_stats, mb / t_stats * 1e3, t_fwd, 2 * mb / t_fwd * 1e3, t_bwd, 5 * mb / t_bwd * 1e3), flush=True)
