#!/usr/bin/env python3
"""Design probe (CPU, numpy): the 8 -> 8 channel 3x3 convolutions of the full-resolution layers (`conv_small_kernel<8>`, the
kernel the bench line's `roofline` names: VALU-bound at ~0.35 of the HBM peak) on 16x16x4 fp32 MFMA tiles with the N
operand packed by a pixel shift.

    y[p][co] = sum_{ky,kx,ci} x[p + (ky-1, kx-1)][ci] * W[ky][kx][ci][co]

As a plain GEMM the tile is 16 pixels x 8 output channels: half of every MFMA is empty and the 72-deep K dimension costs 18
MFMAs per 16 outputs -- exactly the VALU kernel's time (fp32 MFMA = 2x the VALU rate, half wasted).  Pack the empty N half
with the NEXT pixel's outputs: columns 8..15 = y[p + (0,1)][co], which read the same A rows if the K dimension spans kx' in
{-1, 0, 1, 2} (four columns instead of three) and the B operand holds W shifted by one tap in those columns.  K = 3*4*8 = 96
-> 24 MFMAs per 32 outputs = 12 per 16 (x 1.5 fewer); M rows are every second pixel.  This script checks the algebra."""
import numpy as np

rng = np.random.default_rng(1)
H, Wd, CI, CO = 5, 12, 8, 8
x = rng.standard_normal((H, Wd, CI))
w = rng.standard_normal((3, 3, CI, CO))
xp = np.zeros((H + 2, Wd + 3, CI)); xp[1:H + 1, 1:Wd + 1] = x          # zero halo: 1 left / top / bottom, 2 right

want = np.zeros((H, Wd, CO))
for ky in range(3):
    for kx in range(3):
        want += np.einsum("hwi,io->hwo", xp[ky:ky + H, kx:kx + Wd], w[ky, kx])

# B operand [K = (ky, kx', ci)][N = (delta, co)], kx' = 0..3 <-> offsets -1..2
Bm = np.zeros((3, 4, CI, 2, CO))
for ky in range(3):
    for kxp in range(4):
        for d in range(2):
            kx = kxp - d
            if 0 <= kx <= 2:
                Bm[ky, kxp, :, d, :] = w[ky, kx]
Bm = Bm.reshape(3 * 4 * CI, 2 * CO)

got = np.zeros((H, Wd, CO))
tiles = 0
for y in range(H):
    for x0 in range(0, Wd, 2):                                           # one M row per pixel pair
        a = xp[y:y + 3, x0:x0 + 4].reshape(-1)                           # A row: the 3 x 4 x CI patch of pixel (y, x0)
        out = a @ Bm                                                     # [2 * CO]
        got[y, x0] = out[:CO]
        if x0 + 1 < Wd:
            got[y, x0 + 1] = out[CO:]
err = np.abs(got - want).max()
print("K steps per 32 outputs: %d (plain: %d), max |packed - direct| = %.2e" % (3 * 4 * CI // 4, 2 * 3 * 3 * CI // 4, err))
assert err < 1e-12
