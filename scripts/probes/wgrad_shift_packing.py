#!/usr/bin/env python3
"""Design probe (CPU, numpy): weight gradient of an 8 -> 8 channel 3x3 convolution on 16x16 MFMA tiles with BOTH operand
halves packed by pixel shifts.

    dW[ky][kx][co][ci] = sum_p gy[p][co] * x[p + (ky, kx)][ci]          (p over the outputs, x zero-padded, taps -1..1)

Today's packed kernel (conv_wgrad_lds_kernel<..., 8>) puts two taps in the N half-tiles and leaves rows 8..15 of the M
operand empty: 5 MFMA tiles per K step for 9 taps.  If rows 8..15 carry gy shifted by one pixel in x (sa = 1) and the two
N halves carry x shifted by kx = -1 and kx = +1, one tile yields the taps sb - sa = {-1, +1, -2, 0}: three useful taps of
a kernel row, so 3 tiles per K step (0.75 of every tile useful instead of 0.45).  With zero halos on both operands the sums
over the extended pixel range are exact at the borders.  This script checks that algebra against the direct sum."""
import numpy as np

rng = np.random.default_rng(0)
H, W, CO, CI = 6, 9, 8, 8
gy = rng.standard_normal((H, W, CO))
x = rng.standard_normal((H, W, CI))

# direct form
xp = np.zeros((H + 2, W + 2, CI)); xp[1:-1, 1:-1] = x
want = np.zeros((3, 3, CO, CI))
for ky in range(3):
    for kx in range(3):
        want[ky, kx] = np.einsum("hwo,hwi->oi", gy, xp[ky:ky + H, kx:kx + W])

# packed form: K runs over an extended pixel range o = (y, xx) with xx in [-1, W); operands read through zero halos
def g_at(y, xx):          # gy with zero halo
    return gy[y, xx] if 0 <= y < H and 0 <= xx < W else np.zeros(CO)

def x_at(y, xx):
    return x[y, xx] if 0 <= y < H and 0 <= xx < W else np.zeros(CI)

got = np.zeros((3, 3, CO, CI))
tiles = 0
for ky in (-1, 0, 1):
    acc = np.zeros((16, 16))                       # one MFMA accumulator tile, K = every extended pixel
    for y in range(H):
        for xx in range(-1, W):
            a = np.concatenate([g_at(y, xx + sa) for sa in (0, 1)])            # rows: (sa, co)
            b = np.concatenate([x_at(y + ky, xx + sb) for sb in (-1, 1)])      # cols: (sb, ci)
            acc += np.outer(a, b)
    tiles += 1
    for ia, sa in enumerate((0, 1)):
        for ib, sb in enumerate((-1, 1)):
            kx = sb - sa
            if -1 <= kx <= 1:
                blk = acc[ia * 8:(ia + 1) * 8, ib * 8:(ib + 1) * 8]
                # (kx = 0 appears once: sa = 1, sb = +1; kx = -1: sa = 0, sb = -1; kx = +1: sa = 0, sb = +1)
                got[ky + 1, kx + 1] = blk
err = np.abs(got - want).max()
print("tiles per K step: %d (today 5), max |packed - direct| = %.2e" % (tiles, err))
assert err < 1e-12
