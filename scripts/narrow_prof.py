#!/usr/bin/env python3
"""Eager launches of the narrow-layer kernels at the forward's shapes for a rocprofv3 pass (scripts/gpu_narrow_pmc.sh).
Cases are told apart in the profile by kernel name and grid size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from mvster_amd import _lib, ops  # noqa: E402
from scripts.conv_narrow_check import layer_for  # noqa: E402

dev = torch.device("cuda:0")
REP = 6
for cin, B, D, H, W, sk in ((8, 5, 1, 512, 640, False), (8, 5, 1, 512, 640, True), (4, 5, 1, 512, 640, False)):
    layer, _ = layer_for(cin)
    x = torch.randn(B, D, H, W, cin, device=dev)
    skip = torch.randn(B, D, H, W, 8, device=dev) if sk else None
    sm = cp.SKIP_ADD if sk else cp.SKIP_NONE
    for tiles in ((0, 0, 3), (2, 1, 10), (2, 2, 10), (4, 1, 10)):
        for _ in range(REP):
            layer(x, skip=skip, skip_mode=sm, tiles=tiles)
        torch.cuda.synchronize()
L = _lib.load()
for (B, D, Hi, Wi) in ((1, 4, 256, 320), (1, 8, 64, 80)):
    g = torch.Generator().manual_seed(Hi)
    x = torch.randn(B * D, Hi, Wi, 16, generator=g).to(dev)
    w = (torch.randn(3, 3, 16, 8, generator=g) * 0.1).to(dev)
    sc, sh = torch.rand(8, generator=g).to(dev) + 0.5, torch.randn(8, generator=g).to(dev) * 0.1
    skip = torch.randn(B * D, 2 * Hi, 2 * Wi, 8, generator=g).to(dev)
    pw, pb = torch.randn(8, generator=g).to(dev), torch.randn(1, generator=g).to(dev)
    hypo = (500 + 400 * torch.rand(B, D, 2 * Hi, 2 * Wi, generator=g)).to(dev)
    attn = torch.empty_like(hypo)
    outs = [torch.empty(B, 2 * Hi, 2 * Wi, device=dev) for _ in range(4)]
    for _ in range(REP):
        rc = L.mvster_deconv_select(x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), skip.data_ptr(), pw.data_ptr(),
                                    pb.data_ptr(), hypo.data_ptr(), attn.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                    outs[2].data_ptr(), outs[3].data_ptr(), None, B, D, Hi, Wi, 16, 1, 0.5, ops._stream())
        _lib.check(rc, "deconv_select")
    torch.cuda.synchronize()
