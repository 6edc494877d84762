#!/usr/bin/env python3
"""reg2d's last layer + selection in one launch (mvster_deconv_select) at the four stage shapes of the three inference
workloads: isolated hipGraph timings.  Run once with the tree's library and once with MVSTER_LIB=<older build>
MVSTER_LIB_LAX=1 for an A/B.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mvster_amd import _lib, ops  # noqa: E402
from scripts.conv_microbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.load()
print("library:", _lib.LIB_PATH)
for (B, D, Hi, Wi) in ((1, 8, 32, 40), (1, 8, 64, 80), (1, 4, 128, 160), (1, 4, 256, 320), (1, 4, 576, 800), (1, 4, 512, 960), (2, 4, 256, 320)):
    g = torch.Generator().manual_seed(Hi)
    x = torch.randn(B * D, Hi, Wi, 16, generator=g).to(dev)
    w = (torch.randn(3, 3, 16, 8, generator=g) * 0.1).to(dev)
    sc, sh = torch.rand(8, generator=g).to(dev) + 0.5, torch.randn(8, generator=g).to(dev) * 0.1
    skip = torch.randn(B * D, 2 * Hi, 2 * Wi, 8, generator=g).to(dev)
    pw, pb = torch.randn(8, generator=g).to(dev), torch.randn(1, generator=g).to(dev)
    hypo = (500 + 400 * torch.rand(B, D, 2 * Hi, 2 * Wi, generator=g)).to(dev)
    attn = torch.empty_like(hypo)
    outs = [torch.empty(B, 2 * Hi, 2 * Wi, device=dev) for _ in range(4)]

    def run():
        rc = L.mvster_deconv_select(x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), skip.data_ptr(), pw.data_ptr(),
                                    pb.data_ptr(), hypo.data_ptr(), attn.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                    outs[2].data_ptr(), outs[3].data_ptr(), None, B, D, Hi, Wi, 16, 1, 0.5, ops._stream())
        _lib.check(rc, "deconv_select")
    us = min(timeit(run, n=10) for _ in range(3))
    mb = (x.numel() + skip.numel() + 2 * hypo.numel() + 4 * outs[0].numel()) * 4 / 1e6
    print("B%d D%d %4dx%-4d  %7.1f MB  %7.1f us  %5.2f TB/s  %s  checksum %.6e" % (
        B, D, Hi, Wi, mb, us, mb / us, _lib.last_kernel(), attn.double().sum().item() + outs[0].double().sum().item()), flush=True)
