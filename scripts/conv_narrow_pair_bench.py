#!/usr/bin/env python3
"""FPN conv0[0] -> conv0[1] at the forward's shape (5 x 512 x 640): the two launches of the shift-packed kernel against the
one-launch form (mvster_conv_narrow_pair).  Back-to-back launches timed with HIP events, and with a 256 MB flush between
launches (cold L2 / MALL, as inside a forward).
PAIR_DBG=1 with MVSTER_LIB=.../libmvster_hip_probes.so: the knock-out table (parts of the kernel switched off one at a time; the
product library ignores the switches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from mvster_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
NB, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (5, 512, 640)))
g = torch.Generator().manual_seed(1)
a = cp.ConvLayer((torch.randn(8, 3, 1, 3, 3, generator=g) * 0.3).to(dev), False, (1, 1, 1), (0, 1, 1), relu=True, cin_pad=4,
                 bias=torch.randn(8, generator=g).to(dev))
b = cp.ConvLayer((torch.randn(8, 8, 1, 3, 3, generator=g) * 0.2).to(dev), False, (1, 1, 1), (0, 1, 1), relu=True,
                 bias=torch.randn(8, generator=g).to(dev))
x = torch.randn(NB, 1, H, W, 4, generator=g).to(dev)
out = torch.empty(NB, 1, H, W, 8, device=dev)
flush = torch.empty(64 * 1024 * 1024, device=dev)
L = _lib.load()


def two():
    return b(a(x))


def pair(wpc):
    rc = L.mvster_conv_narrow_pair(x.data_ptr(), a.w_small.data_ptr(), a.scale.data_ptr(), a.shift.data_ptr(), b.w_small.data_ptr(),
                                   b.scale.data_ptr(), b.shift.data_ptr(), out.data_ptr(), NB, H, W, 1, 1, wpc, ops._stream())
    _lib.check(rc, "pair")
    return out


def timed(fn, cold, n=30):
    ts = []
    for _ in range(n):
        if cold:
            flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


ref = two()
for wpc in (1, 2):
    got = pair(wpc)
    print("wpc %d: max |pair - two launches| / max = %.2e" % (wpc, (got - ref).abs().max().item() / ref.abs().max().item()),
          "bit-identical" if torch.equal(got, ref) else "")
for cold in (False, True):
    print("%s: two launches %.1f us (%s then %s); one launch wpc 1 %.1f us, wpc 2 %.1f us" % (
        "cold" if cold else "warm", timed(two, cold), "a", "b", timed(lambda: pair(1), cold), timed(lambda: pair(2), cold)))

if os.environ.get("PAIR_DBG"):
    for name, mask in (("all", 0), ("no phase-1 MFMA", 1), ("no phase-2 MFMA", 2), ("no MFMA", 3), ("no stores", 4), ("no loads", 8),
                       ("no memory", 12), ("no MFMA no memory", 15), ("no tiles", 16), ("no epilogues", 32), ("no epilogues no MFMA", 35),
                       ("no epilogues no MFMA no memory", 47), ("no epilogues no memory", 44)):
        print("  %-20s wpc 2: %.1f us   wpc 1: %.1f us" % (name, timed(lambda: pair(2 | mask << 4), True), timed(lambda: pair(1 | mask << 4), True)))
