// Small per-stage kernels of the cascade: camera composition, depth-hypothesis
// schedulers, depth selection, confidence upsampling.  All HBM/latency-bound,
// one thread per output pixel, D (<= 16) handled in registers.
//
// Reference counterparts (models/mvs4net_utils.py):
//   relative projection  :24-26 + :1032-1035      mvster_relative_projection
//   init_inverse_range   :71-77                   mvster_init_range(inverse=1)
//   init_range           :61-69                   mvster_init_range(inverse=0)
//   schedule_inverse_range :79-86                 mvster_schedule_inverse_range
//   schedule_range       :88-99                   mvster_schedule_range
//   prob 1x1x1 + softmax + argmax + gather + confidence + inverse bounds  :900,:1068-1088
//                                                 mvster_select_depth
//   F.interpolate(bilinear, align_corners=True)   :1077   mvster_upsample_bilinear
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int kMaxD = mv::kSelMaxD;
constexpr int kMaxSelectD = 1024;       // select_depth_any_kernel: hypotheses per pixel held in memory, not registers

__global__ void relative_projection_kernel(const float* __restrict__ pm, float* __restrict__ rt, int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int NV = N - 1;
    if (i >= B * NV) return;
    const int b = i / NV, v = i - b * NV;
    const float* ref = pm + ((long)b * N) * 32;
    const float* src = pm + ((long)b * N + v + 1) * 32;
    mv::RT m;
    mv::relative_projection(ref, src, m);
    float* o = rt + (long)i * 12;
    for (int k = 0; k < 9; ++k) o[k] = m.r[k];
    for (int k = 0; k < 3; ++k) o[9 + k] = m.t[k];
}

// depth_values [B, ndv] (first and last column used) -> out [B, D, h, w]
__global__ void init_range_kernel(const float* __restrict__ dv, int ndv, float* __restrict__ out, int B, int D,
                                  int hw, int inverse) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= hw) return;
    mv::init_range_pixel(dv[b * ndv], dv[b * ndv + ndv - 1], out + (long)b * D * hw, D, hw, p, inverse);
}

// inv_min / inv_max [B, h/2, w/2] -> out [B, D, h, w]
__global__ void schedule_inverse_kernel(const float* __restrict__ inv_min, const float* __restrict__ inv_max,
                                        float* __restrict__ out, int B, int D, int h, int w, int hi, int wi) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= h * w) return;
    const long base = (long)b * hi * wi;
    mv::schedule_inverse_pixel(inv_min + base, inv_max + base, out + (long)b * D * h * w, D, h, w, hi, wi, p);
}

// cur_depth [B, h/2, w/2], interval [B] -> out [B, D, h, w]
__global__ void schedule_linear_kernel(const float* __restrict__ cur, const float* __restrict__ interval,
                                       float* __restrict__ out, int B, int D, int h, int w, int hi, int wi) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= h * w) return;
    mv::schedule_linear_pixel(cur + (long)b * hi * wi, interval[b], out + (long)b * D * h * w, D, h, w, hi, wi, p);
}

struct SelectArgs {
    const float* logits;  // [B, D, h, w] or null
    const float* feat;    // [B, D, h, w, CF] channels-last (output of the last reg layer) or null
    const float* prob_w;  // [CF]
    const float* prob_b;  // [1]
    const float* hypo;    // [B, D, h, w]
    float* attn;          // [B, D, h, w]
    float* depth;         // [B, h, w]
    float* conf;          // [B, h, w] or null
    float* inv_min;       // [B, h, w] or null
    float* inv_max;
    float* logits_out;    // optional [B, D, h, w]
    int B, D, hw, CF;
    float split_itv;
};

__global__ void select_depth_kernel(SelectArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= a.hw) return;
    const long vol = (long)b * a.D * a.hw, img = (long)b * a.hw;
    mv::select_pixel(a.logits ? a.logits + vol : nullptr, a.feat ? a.feat + vol * a.CF : nullptr, a.prob_w, a.prob_b,
                     a.CF, a.hypo + vol, a.attn + vol, a.depth + img, a.conf ? a.conf + img : nullptr,
                     a.inv_min ? a.inv_min + img : nullptr, a.inv_max ? a.inv_max + img : nullptr,
                     a.logits_out ? a.logits_out + vol : nullptr, a.D, a.hw, p, a.split_itv);
}

__global__ void select_depth_any_kernel(SelectArgs a) {      // D > kMaxD (mv::select_pixel_any)
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= a.hw) return;
    const long vol = (long)b * a.D * a.hw, img = (long)b * a.hw;
    mv::select_pixel_any(a.logits ? a.logits + vol : nullptr, a.feat ? a.feat + vol * a.CF : nullptr, a.prob_w, a.prob_b,
                         a.CF, a.hypo + vol, a.attn + vol, a.depth + img, a.conf ? a.conf + img : nullptr,
                         a.inv_min ? a.inv_min + img : nullptr, a.inv_max ? a.inv_max + img : nullptr,
                         a.logits_out ? a.logits_out + vol : nullptr, a.D, a.hw, p, a.split_itv);
}

__global__ void upsample_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int hi, int wi,
                                         int ho, int wo) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= ho * wo) return;
    out[(long)b * ho * wo + p] = mv::upsample_pixel(in + (long)b * hi * wi, hi, wi, ho, wo, p);
}

// FPN top-down tail, re-associated (FpnPlan in mvster_amd/conv_plan.py):
//   out4(up(f2) + inner3(c0)) = sum_tap up(W4[tap] f2)[p + tap] + conv3x3(c0; W4*W3) + bias terms
// G = (1x1 conv 64 -> 9*CO of f2 at HALF resolution) is computed by the MFMA kernel; this kernel is
// the bilinear "gather-sum" over the 9 taps:  P[p][co] = sum_{tap inside} ( up(G[..., tap*CO+co])[p+tap] + vb[tap][co] )
// with the reference's x2 align_corners interpolation (mvs4net_utils.py:488) and zero padding (:459).
template <int CO>
__global__ void fpn_tail_gather_kernel(const float* __restrict__ G, const float* __restrict__ vb,
                                       float* __restrict__ P, int NB, int H, int W) {
    const int p = xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const int Hh = H / 2, Wh = W / 2;
    constexpr int CG = 9 * CO;
    const float* g = G + (long)b * Hh * Wh * CG;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int qy = y + ky - 1;
        if (qy < 0 || qy >= H) continue;
        const mv::Lerp ly = mv::make_lerp(qy, Hh, H);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int qx = x + kx - 1;
            if (qx < 0 || qx >= W) continue;
            const mv::Lerp lx = mv::make_lerp(qx, Wh, W);
            const int tap = ky * 3 + kx;
            const float* g00 = g + ((long)ly.i0 * Wh + lx.i0) * CG + tap * CO;
            const float* g01 = g + ((long)ly.i0 * Wh + lx.i1) * CG + tap * CO;
            const float* g10 = g + ((long)ly.i1 * Wh + lx.i0) * CG + tap * CO;
            const float* g11 = g + ((long)ly.i1 * Wh + lx.i1) * CG + tap * CO;
#pragma unroll
            for (int c = 0; c < CO; c += 4) {
                const f32x4 a = ld4(g00 + c), bq = ld4(g01 + c), cq = ld4(g10 + c), dq = ld4(g11 + c);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[c + j] += mv::bilerp(ly, lx, a[j], bq[j], cq[j], dq[j]) + vb[tap * CO + c + j];
            }
        }
    }
    float* o = P + ((long)b * H * W + p) * CO;
#pragma unroll
    for (int c = 0; c < CO; c += 4) {
        f32x4 v = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
        st4(o + c, v);
    }
}

// LDS-tiled form: an 8 x 32 output tile needs only a ~7 x 19 patch of the half-resolution map G (all
// 9*CO channels, <= 46 KB).  The patch is staged once with coalesced 16-byte loads; the 36 bilinear
// corner reads per output pixel then come from LDS instead of 72 L1/L2 gathers per thread.  The kernel is
// VALU-bound, so the interpolation is the flat form  sum_corner (wy*wx) * G  with the 36 corner weights
// formed once per pixel and applied to channel pairs with v_pk_fma_f32 (3x fewer instructions than
// nine bilerp() trees per channel); taps outside the image get zero weights instead of a branch, and the
// bias pushed through the in-bounds taps depends only on the pixel's border class (9 sums, built in LDS).
// COT = total output channels; a workgroup does 8 of them (blockIdx.y picks which), so CO = 16 keeps the same patch size.
template <int COT>
__global__ void __launch_bounds__(256) fpn_tail_gather_lds_kernel(const float* __restrict__ G,
                                                                  const float* __restrict__ vb,
                                                                  float* __restrict__ P, int NB, int H, int W,
                                                                  FastDiv tiles_x, FastDiv tiles_y, float sy, float sx) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int CO = 8;
    constexpr int CGT = 9 * COT;               // channels of G per half-resolution pixel
    constexpr int CG = 9 * CO, Q = CG / 4;     // float4 of them staged per pixel (this workgroup's 8 outputs x 9 taps)
    constexpr int PR = 7, PC = 19;             // patch capacity (rows, cols): 10 output rows x 34 columns at scale < 1/2 touch <= 7 x 19
    //                                            half-resolution pixels -- 38.3 KB of LDS, four workgroups per CU (8 x 20: 46 KB, three)
    const int cbase = blockIdx.y * CO;
    __shared__ f32x4 patch[PR * PC * Q];
    __shared__ float vbsum[9][CO];             // [3*yclass + xclass][c]: sum of vb over the in-bounds taps
    const int Hh = H / 2, Wh = W / 2;
    unsigned txu, tyu;
    const int b = (int)fdivmod(fdivmod(xcd_remap(blockIdx.x, gridDim.x), tiles_x, txu), tiles_y, tyu);
    const int y0 = (int)tyu * 8, x0 = (int)txu * 32;
    // half-resolution footprint of output rows y0-1 .. y0+8 and columns x0-1 .. x0+32 (clamped to the image)
    const int ylo = max(y0 - 1, 0), yhi = min(y0 + 8, H - 1), xlo = max(x0 - 1, 0), xhi = min(x0 + 32, W - 1);
    const int r0 = mv::make_lerp_s(ylo, sy, Hh).i0, r1 = mv::make_lerp_s(yhi, sy, Hh).i1;
    const int c0 = mv::make_lerp_s(xlo, sx, Wh).i0, c1 = mv::make_lerp_s(xhi, sx, Wh).i1;
    const int nr = r1 - r0 + 1, nc = c1 - c0 + 1;       // <= PR, <= PC
    // Staging: ALL of a thread's loads are issued before the first one is consumed (up to 12 x 16 bytes in flight per
    // thread; unused slots fall outside the buffer descriptor and cost nothing).  As a load-store loop the 10-odd trips
    // each waited out a full memory latency, three workgroups per CU could not cover it, and the launch was
    // latency-bound on its own staging (profiles/r03_i_fpn_gather.txt).  Walks the PC-wide patch with compile-time
    // divisors and skips the unused columns.
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(G + (long)b * Hh * Wh * CGT + cbase), (short)0, (int)((long)Hh * Wh * CGT * 4 - cbase * 4), 0x00020000);
    constexpr int NIT = (PR * PC * Q + 255) / 256;
    f32x4 stg[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int q = i % Q, pix = i / Q;
        const int pc = pix % PC, pr = pix / PC;
        const bool ok = pr < nr && pc < nc;
        const unsigned off = ok ? (unsigned)(((r0 + pr) * Wh + (c0 + pc)) * CGT + (q >> 1) * COT + (q & 1) * 4) * 4u : 0xFFFFFFF0u;
        stg[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grsrc, off, 0, 0));
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        if (i < PR * PC * Q) patch[i] = stg[it];               // (i = (pr * PC + pc) * Q + q)
    }
    if (threadIdx.x < 9 * CO) {
        const int cls = threadIdx.x / CO, c = threadIdx.x % CO;
        const int yc = cls / 3, xc = cls % 3;          // 0 = first row/column, 1 = interior, 2 = last
        float sacc = 0.0f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const bool in = !(yc == 0 && ky == 0) && !(yc == 2 && ky == 2) && !(xc == 0 && kx == 0) && !(xc == 2 && kx == 2);
                if (in) sacc += vb[(ky * 3 + kx) * COT + cbase + c];
            }
        vbsum[cls][c] = sacc;
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) return;
    // per-axis taps: row/column offsets into the patch and the two weights (zero outside the image)
    int ry0[3], ry1[3], cx0[3], cx1[3];
    float wy0[3], wy1[3], wx0[3], wx1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int qy = y + k - 1, qx = x + k - 1;
        const bool iy = (unsigned)qy < (unsigned)H, ix = (unsigned)qx < (unsigned)W;
        const mv::Lerp ly = mv::make_lerp_s(iy ? qy : y, sy, Hh), lx = mv::make_lerp_s(ix ? qx : x, sx, Wh);   // (scales formed on the host: six divisions less per thread)
        ry0[k] = (ly.i0 - r0) * PC * Q; ry1[k] = (ly.i1 - r0) * PC * Q;
        cx0[k] = (lx.i0 - c0) * Q;      cx1[k] = (lx.i1 - c0) * Q;
        wy0[k] = iy ? ly.w0 : 0.0f; wy1[k] = iy ? ly.w1 : 0.0f;
        wx0[k] = ix ? lx.w0 : 0.0f; wx1[k] = ix ? lx.w1 : 0.0f;
    }
    const int cls = (y == 0 ? 0 : (y == H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == W - 1 ? 2 : 1));
    f32x2 acc[CO / 2];
#pragma unroll
    for (int c = 0; c < CO / 2; ++c) acc[c] = (f32x2){vbsum[cls][2 * c], vbsum[cls][2 * c + 1]};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tap = ky * 3 + kx;
            const float w00 = wy0[ky] * wx0[kx], w01 = wy0[ky] * wx1[kx], w10 = wy1[ky] * wx0[kx], w11 = wy1[ky] * wx1[kx];
            const f32x4* p00 = patch + ry0[ky] + cx0[kx] + tap * (CO / 4);
            const f32x4* p01 = patch + ry0[ky] + cx1[kx] + tap * (CO / 4);
            const f32x4* p10 = patch + ry1[ky] + cx0[kx] + tap * (CO / 4);
            const f32x4* p11 = patch + ry1[ky] + cx1[kx] + tap * (CO / 4);
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) {
                const f32x4 a = p00[c4], bq = p01[c4], cq = p10[c4], dq = p11[c4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 t = acc[c4 * 2 + h];
                    t += (f32x2){a[2 * h], a[2 * h + 1]} * (f32x2){w00, w00};
                    t += (f32x2){bq[2 * h], bq[2 * h + 1]} * (f32x2){w01, w01};
                    t += (f32x2){cq[2 * h], cq[2 * h + 1]} * (f32x2){w10, w10};
                    t += (f32x2){dq[2 * h], dq[2 * h + 1]} * (f32x2){w11, w11};
                    acc[c4 * 2 + h] = t;
                }
            }
        }
    }
    float* o = P + (((long)b * H + y) * W + x) * COT + cbase;
#pragma unroll
    for (int c = 0; c < CO; c += 4) st4(o + c, (f32x4){acc[c / 2][0], acc[c / 2][1], acc[c / 2 + 1][0], acc[c / 2 + 1][1]});
}

// The finest FPN level's lateral step AND its gather-sum in one launch (round 4): the 72-channel half-resolution map
//   G4[h] = bias + A x[h] + up2(q)[h]        (fpn_lateral_up_kernel: 118 MB written at 5 x 512 x 640 ...)
// only exists to be gathered from (... and 170 MB read back with tile halos by fpn_tail_gather_lds_kernel).  Here a workgroup
// builds the <= 8 x 20 half-resolution patch of G4 that its 8 x 32 output tile reads straight into LDS -- the 16 -> 72
// lateral product on v_mfma_f32_16x16x4_f32 (M = patch pixels, N = 72 channels in five tiles, K = 16 = four steps; D^T form,
// so a lane ends up with four consecutive channels of one pixel), the bilinear x2 of q added in the epilogue -- and then
// runs the gather-sum of fpn_tail_gather_lds_kernel on it unchanged.  HBM traffic: x (26 MB) + q (29 MB) + P (52 MB).
// Same mathematics as the two kernels; the lateral product is summed in the MFMA's K order and up2(q) is the flat
// four-weight FMA form (the kernel is VALU-instruction-bound: 1 120 VALU instructions per wave, PMC), so values agree with
// the two launches to ~3e-7 of the largest output, not bit for bit.
template <int CI, int TH>
__global__ void __launch_bounds__(32 * TH, TH == 8 ? 3 : TH == 16 ? 2 : 1) fpn_tail_fused_kernel(const float* __restrict__ x, const float* __restrict__ A,
                                                             const float* __restrict__ bias, const float* __restrict__ qmap,
                                                             const float* __restrict__ vb, float* __restrict__ P, int NB,
                                                             int H, int W, FastDiv tiles_x, FastDiv tiles_y, float sy, float sx, float sqy, float sqx) {
    static_assert(CI == 16, "one 16-wide K block");
    static_assert(TH == 8 || TH == 16 || TH == 32, "8 x 32 (four waves), 16 x 32 (eight) or 32 x 32 (sixteen) output tiles");
    constexpr int NW = TH / 2, NTHR = 64 * NW;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int CO = 8, CG = 72, Q = CG / 4;
    constexpr int QP = Q + 1;                  // patch pitch in float4 per pixel: 19 (odd) spreads 16 neighbouring pixels over all 16
    //                                            16-byte slot classes of a ds_read_b128 lane group; 18 aliases pixel k with k + 8
    constexpr int PR = TH == 8 ? 7 : TH == 16 ? 11 : 19, PC = 19;   // half-resolution patch of a TH x 32 output tile (+ 1-pixel ring): <= 7 (11) x 19
    constexpr int QR = TH == 8 ? 6 : TH == 16 ? 8 : 11, QC = 11;    // its quarter-resolution footprint: <= 6 (8) x 11
    constexpr int QH = 10;                     // q is staged in two channel halves (quads 0..7, then 8..17) through ONE buffer of
    //                                            10 quads per pixel: 38.3 + 10.6 KB of LDS = three workgroups per CU (two with
    //                                            all 18 quads resident)
    constexpr int MT = (PR * PC + 16 * NW - 1) / (16 * NW);    // M tiles of 16 patch pixels per wave
    extern __shared__ __attribute__((aligned(16))) float fused_lds[];
    f32x4* const patch = reinterpret_cast<f32x4*>(fused_lds);                  // [PR * PC * QP]
    f32x4* const qpatch = patch + PR * PC * QP;                                // [QR * QC * QH]
    float (*const vbsum)[CO] = reinterpret_cast<float (*)[CO]>(qpatch + QR * QC * QH);   // [9][CO]
    const int Hh = H / 2, Wh = W / 2, Hq = Hh / 2, Wq = Wh / 2;
    unsigned txu, tyu;
    const int b = (int)fdivmod(fdivmod(xcd_remap(blockIdx.x, gridDim.x), tiles_x, txu), tiles_y, tyu);
    const int y0 = (int)tyu * TH, x0 = (int)txu * 32;
    const int ylo = max(y0 - 1, 0), yhi = min(y0 + TH, H - 1), xlo = max(x0 - 1, 0), xhi = min(x0 + 32, W - 1);
    const int r0 = mv::make_lerp_s(ylo, sy, Hh).i0, r1 = mv::make_lerp_s(yhi, sy, Hh).i1;
    const int c0 = mv::make_lerp_s(xlo, sx, Wh).i0, c1 = mv::make_lerp_s(xhi, sx, Wh).i1;
    const int nr = r1 - r0 + 1, nc = c1 - c0 + 1;       // <= PR, <= PC
    const int npix = nr * nc;
    // quarter-resolution footprint of the patch
    const int qr0 = mv::make_lerp_s(r0, sqy, Hq).i0, qr1 = mv::make_lerp_s(r1, sqy, Hq).i1;
    const int qc0 = mv::make_lerp_s(c0, sqx, Wq).i0, qc1 = mv::make_lerp_s(c1, sqx, Wq).i1;
    const int qnr = qr1 - qr0 + 1, qnc = qc1 - qc0 + 1;  // <= QR, <= QC

    // ---- phase A: the lateral step of this tile's patch, into LDS ---------------------------------------------------------
    // every global load of the phase is in flight before the first one is consumed: the quarter-resolution patch of q (staged
    // in LDS: each of its values is a corner of ~9 patch pixels) and this lane's input pixels (the MFMA B operand)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lm = lane & 15, lq = lane >> 4;
    {
        const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(qmap + (long)b * Hq * Wq * CG), (short)0, (int)((long)Hq * Wq * CG * 4), 0x00020000);
        constexpr int NIT0 = (QR * QC * 8 + NTHR - 1) / NTHR, NIT1 = (QR * QC * 10 + NTHR - 1) / NTHR;
        f32x4 stg0[NIT0], stg1[NIT1];
        auto stage_load = [&](int i, int nq, int q0) -> f32x4 {             // slot i of a half with nq quads per pixel from quad q0
            const int qd = i % nq, pix = i / nq;
            const int pc = pix % QC, pr = pix / QC;
            const bool ok = pr < qnr && pc < qnc;
            const unsigned off = ok ? (unsigned)(((qr0 + pr) * Wq + (qc0 + pc)) * CG + (q0 + qd) * 4) * 4u : 0xFFFFFFF0u;
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrsrc, off, 0, 0));
        };
#pragma unroll
        for (int it = 0; it < NIT0; ++it) stg0[it] = stage_load(threadIdx.x + it * NTHR, 8, 0);
#pragma unroll
        for (int it = 0; it < NIT1; ++it) stg1[it] = stage_load(threadIdx.x + it * NTHR, 10, 8);
        // this lane's patch pixels (M tile NW mt + wave: pixels (NW mt + wave) * 16 + lm) and their input vectors
        const float* xb = x + (long)b * Hh * Wh * CI;
        const FastDiv ncd = mv_fastdiv_dev((unsigned)nc);
        f32x4 xv[MT];
        int ppos[MT];                                   // pr | pc << 8, -1 = no pixel
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int i = (NW * mt + wave) * 16 + lm;
            const bool valid = i < npix;
            unsigned pcu;
            const int pr = (int)fdivmod((unsigned)(valid ? i : 0), ncd, pcu), pc = (int)pcu;
            ppos[mt] = valid ? (pr | (pc << 8)) : -1;
            xv[mt] = ld4(xb + ((long)(r0 + pr) * Wh + (c0 + pc)) * CI + 4 * lq);
        }
        // weights as the A operand: row lm of N tile nt = channel nt * 16 + lm, K slot (j, lq) = input channel 4 lq + j
        float aw[5][4];
        f32x4 bv[5];
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
            const int co = nt * 16 + lm;
            const f32x4 v = co < CG ? ld4(A + co * CI + 4 * lq) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) aw[nt][j] = v[j];
            bv[nt] = ld4(bias + (nt * 4 + lq < Q ? nt * 16 + 4 * lq : 0));
        }
        // one half: N tiles NT0 .. NT1 - 1 of every M tile of this wave, q quads from QBASE in the buffer
        auto half = [&](auto nt0c, auto nt1c, auto qbasec) {
            constexpr int NT0 = decltype(nt0c)::value, NT1 = decltype(nt1c)::value, QBASE = decltype(qbasec)::value;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if ((NW * mt + wave) * 16 >= npix) break;           // (wave-uniform)
                const bool valid = ppos[mt] >= 0;
                const int pr = valid ? ppos[mt] & 255 : 0, pc = valid ? ppos[mt] >> 8 : 0;
                const mv::Lerp ly = mv::make_lerp_s(r0 + pr, sqy, Hq), lx = mv::make_lerp_s(c0 + pc, sqx, Wq);
                // (flat 4-weight form on FMAs: 4 instead of 10 packed operations per channel pair)
                const float w00 = ly.w0 * lx.w0, w01 = ly.w0 * lx.w1, w10 = ly.w1 * lx.w0, w11 = ly.w1 * lx.w1;
                const f32x4* q00p = qpatch + ((ly.i0 - qr0) * QC + (lx.i0 - qc0)) * QH + lq - QBASE;
                const f32x4* q01p = qpatch + ((ly.i0 - qr0) * QC + (lx.i1 - qc0)) * QH + lq - QBASE;
                const f32x4* q10p = qpatch + ((ly.i1 - qr0) * QC + (lx.i0 - qc0)) * QH + lq - QBASE;
                const f32x4* q11p = qpatch + ((ly.i1 - qr0) * QC + (lx.i1 - qc0)) * QH + lq - QBASE;
                f32x4 acc[NT1 - NT0];
#pragma unroll
                for (int nt = NT0; nt < NT1; ++nt) acc[nt - NT0] = bv[nt];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = NT0; nt < NT1; ++nt)
                        acc[nt - NT0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[nt][j], xv[mt][j], acc[nt - NT0], 0, 0, 0);
#pragma unroll
                for (int nt = NT0; nt < NT1; ++nt) {
                    if (valid && nt * 4 + lq < Q) {                // (the fifth N tile holds channels 64..71 only)
                        const f32x4 a00 = q00p[nt * 4], a01 = q01p[nt * 4], a10 = q10p[nt * 4], a11 = q11p[nt * 4];
                        f32x4 r;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            r[k] = fmaf(w00, a00[k], fmaf(w01, a01[k], fmaf(w10, a10[k], fmaf(w11, a11[k], acc[nt - NT0][k]))));
                        patch[(pr * PC + pc) * QP + nt * 4 + lq] = r;
                    }
                }
            }
        };
#pragma unroll
        for (int it = 0; it < NIT0; ++it) {
            const int i = threadIdx.x + it * NTHR;
            if (i < QR * QC * 8) qpatch[(i / 8) * QH + (i % 8)] = stg0[it];
        }
        __syncthreads();
        half(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});      // channels 0..31
        __syncthreads();                                           // everyone is done with the first half of q
#pragma unroll
        for (int it = 0; it < NIT1; ++it) {
            const int i = threadIdx.x + it * NTHR;
            if (i < QR * QC * 10) qpatch[(i / 10) * QH + (i % 10)] = stg1[it];
        }
        __syncthreads();
        half(std::integral_constant<int, 2>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 8>{});      // channels 32..71
    }
    if (threadIdx.x < 9 * CO) {
        const int cls = threadIdx.x / CO, c = threadIdx.x % CO;
        const int yc = cls / 3, xc = cls % 3;          // 0 = first row/column, 1 = interior, 2 = last
        float sacc = 0.0f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const bool in = !(yc == 0 && ky == 0) && !(yc == 2 && ky == 2) && !(xc == 0 && kx == 0) && !(xc == 2 && kx == 2);
                if (in) sacc += vb[(ky * 3 + kx) * CO + c];
            }
        vbsum[cls][c] = sacc;
    }
    __syncthreads();

    // ---- phase B: the gather-sum (fpn_tail_gather_lds_kernel<8>, verbatim) ----------------------------------------------
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int y = y0 + ty, xo = x0 + tx;
    if (y >= H || xo >= W) return;
    int ry0[3], ry1[3], cx0[3], cx1[3];
    float wy0[3], wy1[3], wx0[3], wx1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int qy = y + k - 1, qx = xo + k - 1;
        const bool iy = (unsigned)qy < (unsigned)H, ix = (unsigned)qx < (unsigned)W;
        const mv::Lerp ly = mv::make_lerp_s(iy ? qy : y, sy, Hh), lx = mv::make_lerp_s(ix ? qx : xo, sx, Wh);
        ry0[k] = (ly.i0 - r0) * PC * QP; ry1[k] = (ly.i1 - r0) * PC * QP;
        cx0[k] = (lx.i0 - c0) * QP;     cx1[k] = (lx.i1 - c0) * QP;
        wy0[k] = iy ? ly.w0 : 0.0f; wy1[k] = iy ? ly.w1 : 0.0f;
        wx0[k] = ix ? lx.w0 : 0.0f; wx1[k] = ix ? lx.w1 : 0.0f;
    }
    const int cls = (y == 0 ? 0 : (y == H - 1 ? 2 : 1)) * 3 + (xo == 0 ? 0 : (xo == W - 1 ? 2 : 1));
    f32x2 acc[CO / 2];
#pragma unroll
    for (int c = 0; c < CO / 2; ++c) acc[c] = (f32x2){vbsum[cls][2 * c], vbsum[cls][2 * c + 1]};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tap = ky * 3 + kx;
            const float w00 = wy0[ky] * wx0[kx], w01 = wy0[ky] * wx1[kx], w10 = wy1[ky] * wx0[kx], w11 = wy1[ky] * wx1[kx];
            const f32x4* p00 = patch + ry0[ky] + cx0[kx] + tap * (CO / 4);
            const f32x4* p01 = patch + ry0[ky] + cx1[kx] + tap * (CO / 4);
            const f32x4* p10 = patch + ry1[ky] + cx0[kx] + tap * (CO / 4);
            const f32x4* p11 = patch + ry1[ky] + cx1[kx] + tap * (CO / 4);
#pragma unroll
            for (int c4 = 0; c4 < CO / 4; ++c4) {
                const f32x4 a = p00[c4], bq = p01[c4], cq = p10[c4], dq = p11[c4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 t = acc[c4 * 2 + h];
                    t += (f32x2){a[2 * h], a[2 * h + 1]} * (f32x2){w00, w00};
                    t += (f32x2){bq[2 * h], bq[2 * h + 1]} * (f32x2){w01, w01};
                    t += (f32x2){cq[2 * h], cq[2 * h + 1]} * (f32x2){w10, w10};
                    t += (f32x2){dq[2 * h], dq[2 * h + 1]} * (f32x2){w11, w11};
                    acc[c4 * 2 + h] = t;
                }
            }
        }
    }
    float* o = P + (((long)b * H + y) * W + xo) * CO;
#pragma unroll
    for (int c = 0; c < CO; c += 4) st4(o + c, (f32x4){acc[c / 2][0], acc[c / 2][1], acc[c / 2 + 1][0], acc[c / 2 + 1][1]});
}

// Adjoint of the gather-sum above with respect to G (training): gG [NB, H/2, W/2, pitch] <- gP [NB, H, W, CO],
//   gG[q][tap*CO + co] = sum over full-resolution positions r = p + tap (both p and r inside the image) of
//                        w(r -> q) * gP[p][co],     w = the bilinear x2 weight with which r reads q,
// as a gather over the <= 6 x 6 positions r that can touch q (same make_lerp() as the forward; no atomics).  One
// thread per (q, tap): the nine taps of a pixel write one contiguous run.  Channels 9*CO .. pitch-1 are zeroed (the
// input-gradient convolution that follows wants a multiple of 16 channels).
template <int CO>
__global__ void __launch_bounds__(256) fpn_tail_gather_bwd_kernel(const float* __restrict__ gP, float* __restrict__ gG,
                                                                  int NB, int H, int W, int pitch, FastDiv wdiv, FastDiv hdiv) {
    const int Hh = H / 2, Wh = W / 2;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)NB * Hh * Wh * 9) return;
    const int tap = (int)(i % 9);
    const unsigned q = i / 9;
    unsigned xu, yu;
    const int b = (int)fdivmod(fdivmod(q, wdiv, xu), hdiv, yu);
    const int xi = (int)xu, yi = (int)yu;
    const int ty = tap / 3 - 1, tx = tap % 3 - 1;
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int ro = 2 * yi - 2 + k, co = 2 * xi - 2 + k;
        const bool vy = ro >= 0 && ro < H && ro - ty >= 0 && ro - ty < H;
        const bool vx = co >= 0 && co < W && co - tx >= 0 && co - tx < W;
        const mv::Lerp ly = mv::make_lerp(vy ? ro : 0, Hh, H), lx = mv::make_lerp(vx ? co : 0, Wh, W);
        wy[k] = vy ? (ly.i0 == yi ? ly.w0 : 0.0f) + (ly.i1 == yi ? ly.w1 : 0.0f) : 0.0f;
        wx[k] = vx ? (lx.i0 == xi ? lx.w0 : 0.0f) + (lx.i1 == xi ? lx.w1 : 0.0f) : 0.0f;
    }
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
    const float* base = gP + (long)b * H * W * CO;
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
        if (wy[ky] == 0.0f) continue;
        const int py = 2 * yi - 2 + ky - ty;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            if (wx[kx] == 0.0f) continue;
            const int px = 2 * xi - 2 + kx - tx;
            const float w = wy[ky] * wx[kx];
            const float* g = base + ((long)py * W + px) * CO;
#pragma unroll
            for (int c = 0; c < CO; c += 4) {
                const f32x4 v = ld4(g + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[c + j] = fmaf(w, v[j], acc[c + j]);
            }
        }
    }
    float* o = gG + (long)q * pitch + tap * CO;
#pragma unroll
    for (int c = 0; c < CO; c += 4) st4(o + c, (f32x4){acc[c], acc[c + 1], acc[c + 2], acc[c + 3]});
    if (tap == 0)
        for (int c = 9 * CO; c < pitch; c += 4) st4(gG + (long)q * pitch + c, (f32x4){0.f, 0.f, 0.f, 0.f});
}

// The same adjoint from an LDS tile (round 6): a workgroup of 128 threads owns 8 x 16 half-resolution pixels q, stages the
// 22 x 38 full-resolution pixels of gP their nine taps can reach (27 KB, zero outside the image) and every thread walks ONE q
// with all nine taps: the 8 x 8 window of gP around q is read once from LDS instead of 9 x 16 times from L1 / L2 (3.8 GB of
// 16-byte loads per launch at 10 x 512 x 640 in the form above: 250 us against 60-90 us of HBM time).  The 9 x 8 sums of a
// thread leave through LDS too, so that the stores of a wavefront are whole rows of the output (a thread's own 320 bytes
// would be 20 partial-line stores).  Same weights as fpn_tail_gather_bwd_kernel, summed separably (rows, then columns).
constexpr int kGbTQY = 8, kGbTQX = 16, kGbPH = 2 * kGbTQY + 6, kGbPW = 2 * kGbTQX + 6, kGbThreads = kGbTQY * kGbTQX;

template <int CO>
__global__ void __launch_bounds__(kGbThreads) fpn_tail_gather_bwd_lds_kernel(const float* __restrict__ gP, float* __restrict__ gG,
                                                                             int NB, int H, int W, int pitch, int tiles_x,
                                                                             int tiles_y) {
    static_assert(CO == 8, "8 output channels (the finest level)");
    constexpr int kIn = kGbPH * kGbPW * (CO / 4), kOutMax = kGbThreads * 20;          // float4: input tile / output staging
    __shared__ f32x4 smem[kIn > kOutMax ? kIn : kOutMax];
    const int Hh = H / 2, Wh = W / 2;
    int t = blockIdx.x;
    const int tx_ = t % tiles_x;
    t /= tiles_x;
    const int ty_ = t % tiles_y;
    const int b = t / tiles_y;
    const int qy0 = ty_ * kGbTQY, qx0 = tx_ * kGbTQX;
    const int py0 = 2 * qy0 - 3, px0 = 2 * qx0 - 3;
    const float* base = gP + (long)b * H * W * CO;
    for (int i = threadIdx.x; i < kIn; i += kGbThreads) {
        const int c4 = i % (CO / 4), pix = i / (CO / 4);
        const int col = pix % kGbPW, row = pix / kGbPW;
        const int py = py0 + row, px = px0 + col;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (py >= 0 && py < H && px >= 0 && px < W) v = ld4(base + ((long)py * W + px) * CO + c4 * 4);
        smem[i] = v;
    }
    __syncthreads();
    const int lqx = threadIdx.x % kGbTQX, lqy = threadIdx.x / kGbTQX;
    const int xi = qx0 + lqx, yi = qy0 + lqy;
    const bool active = xi < Wh && yi < Hh;
    // weight with which full-resolution position r = 2q - 2 + k reads q (0 outside the image)
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int ro = 2 * yi - 2 + k, co = 2 * xi - 2 + k;
        const bool vy = active && ro >= 0 && ro < H, vx = active && co >= 0 && co < W;
        const mv::Lerp ly = mv::make_lerp(vy ? ro : 0, Hh, H), lx = mv::make_lerp(vx ? co : 0, Wh, W);
        wy[k] = vy ? (ly.i0 == yi ? ly.w0 : 0.0f) + (ly.i1 == yi ? ly.w1 : 0.0f) : 0.0f;
        wx[k] = vx ? (lx.i0 == xi ? lx.w0 : 0.0f) + (lx.i1 == xi ? lx.w1 : 0.0f) : 0.0f;
    }
    float acc[9][CO];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[tp][c] = 0.0f;
    // p = 2q - 3 + (a, bx); tap (ty, tx) in {-1, 0, 1}^2 reads it at r = p + tap, i.e. k = a - 1 + ty, kk = bx - 1 + tx.
    // Separable: per window row a, T[tx][c] = sum_bx wx[bx - 1 + tx] gP[a][bx][c], then acc[ty][tx] += wy[a - 1 + ty] T[tx]
    // -- no per-lane tests (zero weights and the tile's zero border do the masking), 216 instead of ~450 instructions per
    // row.  (Another association than the form above: the results agree to rounding, not to the bit.)
    // (a real loop over the rows: fully unrolled, the scheduler hoists all 128 LDS reads of the window -- 418 VGPRs, one wave
    //  per SIMD; the row's three weights are picked with selects instead of a dynamically indexed array)
    auto wy_at = [&](int k) -> float {
        float w = 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) w = k == i ? wy[i] : w;
        return w;
    };
#pragma unroll 1
    for (int a = 0; a < 8; ++a) {
        const int prow = 2 * lqy + a;
        float T[3][CO];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
            for (int c = 0; c < CO; ++c) T[tx][c] = 0.0f;
#pragma unroll
        for (int bx = 0; bx < 8; ++bx) {
            const int pcol = 2 * lqx + bx;
            const f32x4 v0 = smem[(prow * kGbPW + pcol) * 2], v1 = smem[(prow * kGbPW + pcol) * 2 + 1];
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int kk = bx - 2 + tx;                 // (tx here = offset + 1)
                if (kk < 0 || kk >= 6) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    T[tx][j] = fmaf(wx[kk], v0[j], T[tx][j]);
                    T[tx][4 + j] = fmaf(wx[kk], v1[j], T[tx][4 + j]);
                }
            }
        }
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const float wk = wy_at(a - 2 + ty);             // (0 outside 0..5)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int c = 0; c < CO; ++c) acc[ty * 3 + tx][c] = fmaf(wk, T[tx][c], acc[ty * 3 + tx][c]);
        }
    }
    __syncthreads();                                       // (every thread is done with the input tile)
    const int p4 = pitch >> 2;                             // float4 per output pixel (18 or 20)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        smem[threadIdx.x * p4 + tp * 2] = (f32x4){acc[tp][0], acc[tp][1], acc[tp][2], acc[tp][3]};
        smem[threadIdx.x * p4 + tp * 2 + 1] = (f32x4){acc[tp][4], acc[tp][5], acc[tp][6], acc[tp][7]};
    }
    for (int c = 18; c < p4; ++c) smem[threadIdx.x * p4 + c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    // a tile row = kGbTQX pixels x pitch floats, contiguous in gG
    const int rowlen = kGbTQX * p4;
    for (int i = threadIdx.x; i < kGbTQY * rowlen; i += kGbThreads) {
        const int r = i / rowlen, j = i - r * rowlen;
        const int qx = qx0 + j / p4, qy = qy0 + r;
        if (qy < Hh && qx < Wh) st4(gG + (((long)b * Hh + qy) * Wh + qx0) * pitch + (long)j * 4, smem[i]);
    }
}

// Lateral 1x1 conv + top-down add of the FPN (models/mvs4net_utils.py:485) for a top-down map that only exists at
// the coarser level:   out[p][co] = bias[co] + sum_ci A[co][ci] x[p][ci] + up2(q)[p][co]
// with x [NB,H,W,CI] the bottom-up map, q [NB,H/2,W/2,CO] and up2 = the reference's x2 align_corners interpolation.
// K = CI = 16 is one MFMA step, so the GEMM form is all epilogue (4-byte corner gathers per accumulator element:
// 127 us at 5 x 256 x 320 x 72), and a thread-per-pixel form reads and writes 16-byte slices 288 bytes apart, which
// thrashes the 32 KB L1 (122 us).  Here a thread owns one float4 of output channels (its 4 x CI weights stay in
// registers) and consecutive threads consecutive channels, so every corner read and every store of a wavefront is a
// contiguous run of whole cache lines; a workgroup of (CO/4) x 14 threads walks 14 pixels at a time.
constexpr int kLateralSlots = 14, kLateralIters = 16;   // 18 x 14 = 252 threads

template <int CI, int CO>
__global__ void __launch_bounds__(CO / 4 * kLateralSlots)
fpn_lateral_up_kernel(const float* __restrict__ x, const float* __restrict__ A, const float* __restrict__ bias,
                      const float* __restrict__ q, float* __restrict__ out, int H, int W, FastDiv wdiv) {
    constexpr int NCH = CO / 4;
    const int chunk = threadIdx.x % NCH, slot = threadIdx.x / NCH;
    const int b = blockIdx.y;
    const int Hh = H / 2, Wh = W / 2;
    float a[4][CI];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CI; c += 4) {
            const f32x4 v = ld4(A + (chunk * 4 + j) * CI + c);
            a[j][c] = v[0]; a[j][c + 1] = v[1]; a[j][c + 2] = v[2]; a[j][c + 3] = v[3];
        }
    const f32x4 bv = ld4(bias + chunk * 4);
    const float* qb = q + (long)b * Hh * Wh * CO + chunk * 4;
    const int base = xcd_remap(blockIdx.x, gridDim.x) * (kLateralSlots * kLateralIters);
    // Software-pipelined over the workgroup's 16 pixels per slot: the nine loads of pixel it + 1 are in flight under the
    // arithmetic of pixel it.  (As load -> compute -> store trips each trip waited out a memory latency; issuing the loads
    // of 2 or 4 pixels together and then computing them was measured no better: 54 / 59 / 59 us.)
    struct Px {
        mv::Lerp ly, lx;
        f32x4 xv[CI / 4], q00, q01, q10, q11;
        int p;
    };
    auto fetch = [&](int it, Px& o) {
        const int p = base + it * kLateralSlots + slot;
        o.p = p;
        const int pc = min(p, H * W - 1);
        unsigned xu;
        const int y = (int)fdivmod((unsigned)pc, wdiv, xu);
        const int xx = (int)xu;
        o.ly = mv::make_lerp(y, Hh, H);
        o.lx = mv::make_lerp(xx, Wh, W);
        const float* xp = x + ((long)b * H * W + pc) * CI;
#pragma unroll
        for (int c = 0; c < CI / 4; ++c) o.xv[c] = ld4(xp + 4 * c);
        o.q00 = ld4(qb + ((long)o.ly.i0 * Wh + o.lx.i0) * CO);
        o.q01 = ld4(qb + ((long)o.ly.i0 * Wh + o.lx.i1) * CO);
        o.q10 = ld4(qb + ((long)o.ly.i1 * Wh + o.lx.i0) * CO);
        o.q11 = ld4(qb + ((long)o.ly.i1 * Wh + o.lx.i1) * CO);
    };
    auto finish = [&](const Px& o) {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = bv[j];
#pragma unroll
            for (int c = 0; c < CI; ++c) acc = fmaf(a[j][c], o.xv[c / 4][c % 4], acc);
            r[j] = mv::bilerp(o.ly, o.lx, o.q00[j], o.q01[j], o.q10[j], o.q11[j]) + acc;
        }
        if (o.p < H * W) st4(out + ((long)b * H * W + o.p) * CO + chunk * 4, r);
    };
    Px cur, nxt;
    fetch(0, cur);
#pragma unroll 1
    for (int it = 0; it < kLateralIters; it += 2) {
        fetch(it + 1, nxt);
        finish(cur);
        fetch(it + 2 < kLateralIters ? it + 2 : it + 1, cur);      // (unconditional: the last trip re-reads its pixel)
        finish(nxt);
    }
}

// Separable form of fpn_tail_gather (2.4x fewer loads): bilinear interpolation factorises into a
// vertical and a horizontal 1-D lerp.
//   pass 1  V[b][y][xh][kx*CO + co] = sum_{ky inside} lerp_y(y+ky-1)( G[b][.][xh][(ky*3+kx)*CO + co] )
//   pass 2  P[b][y][x][co] = sum_{kx inside} ( lerp_x(x+kx-1)( V[b][y][.][kx*CO + co] ) + sum_{ky inside} vb[ky*3+kx][co] )
template <int CO>
__global__ void fpn_tail_vpass_kernel(const float* __restrict__ G, float* __restrict__ V, int NB, int H, int W) {
    const int Hh = H / 2, Wh = W / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (y, xh)
    const int b = blockIdx.y;
    if (i >= H * Wh) return;
    const int y = i / Wh, xh = i - y * Wh;
    constexpr int CG = 9 * CO, CV = 3 * CO;
    const float* g = G + ((long)b * Hh * Wh + xh) * CG;
    float acc[CV];
#pragma unroll
    for (int c = 0; c < CV; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int qy = y + ky - 1;
        if (qy < 0 || qy >= H) continue;
        const mv::Lerp ly = mv::make_lerp(qy, Hh, H);
        const float* r0 = g + (long)ly.i0 * Wh * CG + ky * CV;
        const float* r1 = g + (long)ly.i1 * Wh * CG + ky * CV;
#pragma unroll
        for (int c = 0; c < CV; c += 4) {
            const f32x4 a = ld4(r0 + c), bq = ld4(r1 + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c + j] += ly.w0 * a[j] + ly.w1 * bq[j];
        }
    }
    float* o = V + ((long)b * H * Wh + i) * CV;
#pragma unroll
    for (int c = 0; c < CV; c += 4) st4(o + c, (f32x4){acc[c], acc[c + 1], acc[c + 2], acc[c + 3]});
}

template <int CO>
__global__ void fpn_tail_hpass_kernel(const float* __restrict__ V, const float* __restrict__ vb,
                                      float* __restrict__ P, int NB, int H, int W) {
    const int Wh = W / 2;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    constexpr int CV = 3 * CO;
    const float* v = V + ((long)b * H + y) * Wh * CV;
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int qx = x + kx - 1;
        if (qx < 0 || qx >= W) continue;
        const mv::Lerp lx = mv::make_lerp(qx, Wh, W);
        const float* c0 = v + (long)lx.i0 * CV + kx * CO;
        const float* c1 = v + (long)lx.i1 * CV + kx * CO;
#pragma unroll
        for (int c = 0; c < CO; c += 4) {
            const f32x4 a = ld4(c0 + c), bq = ld4(c1 + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c + j] += lx.w0 * a[j] + lx.w1 * bq[j];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int qy = y + ky - 1;
            if (qy < 0 || qy >= H) continue;
#pragma unroll
            for (int c = 0; c < CO; ++c) acc[c] += vb[(ky * 3 + kx) * CO + c];
        }
    }
    float* o = P + ((long)b * H * W + p) * CO;
#pragma unroll
    for (int c = 0; c < CO; c += 4) st4(o + c, (f32x4){acc[c], acc[c + 1], acc[c + 2], acc[c + 3]});
}

struct PackArgs {
    const float* img[16];   // N views, each [B,3,H,W]
    float* out;             // [N*B, H, W, 4]  (RGB0, channels-last)
    int N, B, HW;
};

// list of N [B,3,H,W] images -> one channels-last RGB0 batch (feeds FPN4 conv0, mvs4net_utils.py:427)
__global__ void pack_images_kernel(PackArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int vb = blockIdx.y;            // v * B + b
    if (p >= a.HW) return;
    const int v = vb / a.B, b = vb - v * a.B;
    const float* src = a.img[v] + (long)b * 3 * a.HW + p;
    const f32x4 px = {src[0], src[a.HW], src[2 * (long)a.HW], 0.0f};
    st4(a.out + ((long)vb * a.HW + p) * 4, px);
}

struct MultiProjArgs {
    const float* pm[8];     // per stage [B,N,2,4,4]
    float* rt;              // [nstage, B, N-1, 12]
    int nstage, B, N;
};

__global__ void relative_projection_multi_kernel(MultiProjArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int NV = a.N - 1;
    const int per = a.B * NV;
    if (i >= a.nstage * per) return;
    const int s = i / per, r = i - s * per;
    const int b = r / NV, v = r - b * NV;
    const float* ref = a.pm[s] + ((long)b * a.N) * 32;
    const float* src = a.pm[s] + ((long)b * a.N + v + 1) * 32;
    mv::RT m;
    mv::relative_projection(ref, src, m);
    float* o = a.rt + (long)i * 12;
    for (int k = 0; k < 9; ++k) o[k] = m.r[k];
    for (int k = 0; k < 3; ++k) o[9 + k] = m.t[k];
}

// The three launches every forward starts with, in one: pack the views, the first stage's hypotheses (mvs4net_utils.py:61-86 on
// the [first, last] depth of depth_values) and every stage's relative projections (first workgroups of their grid row).  Same
// device functions as the separate kernels: the same bits.
struct PrologueArgs {
    PackArgs pack;
    MultiProjArgs proj;
    const float* dv;        // [B, ndv]
    float* hypo;            // [B, D, h, w]
    int ndv, D, hw, inverse;
};

__global__ void __launch_bounds__(256) forward_prologue_kernel(PrologueArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    // rows 0 and 1 of the grid are the side jobs (dispatched first: the serial 4x4 fp64 inverses take ~7 us and must not
    // start after the pack has drained), rows 2.. the views
    const int NB = a.pack.N * a.pack.B;
    const int vb = blockIdx.y == 0 ? NB + 1 : blockIdx.y == 1 ? NB : (int)blockIdx.y - 2;
    if (vb < NB) {
        if (p >= a.pack.HW) return;
        const int v = vb / a.pack.B, b = vb - v * a.pack.B;
        const float* src = a.pack.img[v] + (long)b * 3 * a.pack.HW + p;
        const f32x4 px = {src[0], src[a.pack.HW], src[2 * (long)a.pack.HW], 0.0f};
        st4(a.pack.out + ((long)vb * a.pack.HW + p) * 4, px);
    } else if (vb == NB) {
        if (p >= a.hw) return;
        for (int b = 0; b < a.pack.B; ++b)
            mv::init_range_pixel(a.dv[b * a.ndv], a.dv[b * a.ndv + a.ndv - 1], a.hypo + (long)b * a.D * a.hw, a.D, a.hw, p,
                                 a.inverse);
    } else {
        const int NV = a.proj.N - 1;
        const int per = a.proj.B * NV;
        if (p >= a.proj.nstage * per) return;
        const int s = p / per, r = p - s * per;
        const int b = r / NV, v = r - b * NV;
        mv::RT m;
        mv::relative_projection(a.proj.pm[s] + ((long)b * a.proj.N) * 32, a.proj.pm[s] + ((long)b * a.proj.N + v + 1) * 32, m);
        float* o = a.proj.rt + (long)p * 12;
        for (int k = 0; k < 9; ++k) o[k] = m.r[k];
        for (int k = 0; k < 3; ++k) o[9 + k] = m.t[k];
    }
}

struct MultiUpArgs {
    const float* in[8];     // [B, hi[k], wi[k]]
    float* out[8];          // [B, ho, wo]
    int hi[8], wi[8];
    int ho, wo;
};

// several maps of one batch to one output size in one launch (the coarse stages' confidence maps, MVS4Net.py:1077 per stage)
__global__ void __launch_bounds__(256) upsample_bilinear_multi_kernel(MultiUpArgs a) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y, k = blockIdx.z;
    if (p >= a.ho * a.wo) return;
    a.out[k][(long)b * a.ho * a.wo + p] = mv::upsample_pixel(a.in[k] + (long)b * a.hi[k] * a.wi[k], a.hi[k], a.wi[k], a.ho, a.wo, p);
}

}  // namespace

extern "C" int mvster_relative_projection(const float* proj_matrices, float* rt, int B, int N, void* stream) {
    if (!proj_matrices || !rt) return MVSTER_ERR_NULL;
    if (B <= 0 || N < 2) return MVSTER_ERR_SHAPE;
    const int n = B * (N - 1);
    hipLaunchKernelGGL(relative_projection_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       proj_matrices, rt, B, N);
    return mv_check_launch();
}

extern "C" int mvster_init_range(const float* depth_values, int ndv, float* out, int B, int D, int h, int w,
                                 int inverse, void* stream) {
    if (!depth_values || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || D < 2 || h <= 0 || w <= 0 || ndv < 1) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(init_range_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       depth_values, ndv, out, B, D, h * w, inverse);
    return mv_check_launch();
}

extern "C" int mvster_schedule_inverse_range(const float* inv_min, const float* inv_max, float* out, int B, int D,
                                             int h, int w, void* stream) {
    if (!inv_min || !inv_max || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || D < 2 || h < 2 || w < 2) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(schedule_inverse_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       inv_min, inv_max, out, B, D, h, w, h / 2, w / 2);
    return mv_check_launch();
}

extern "C" int mvster_schedule_range(const float* cur_depth, const float* interval, float* out, int B, int D, int h,
                                     int w, void* stream) {
    if (!cur_depth || !interval || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || D < 2 || h < 2 || w < 2) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(schedule_linear_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       cur_depth, interval, out, B, D, h, w, h / 2, w / 2);
    return mv_check_launch();
}

extern "C" int mvster_select_depth(const float* logits, const float* feat, const float* prob_w, const float* prob_b,
                                   int CF, const float* hypo, float* attn, float* depth, float* conf, float* inv_min,
                                   float* inv_max, float* logits_out, int B, int D, int h, int w, float split_itv,
                                   void* stream) {
    if ((!logits && !feat) || !hypo || !attn || !depth) return MVSTER_ERR_NULL;
    if (feat && (!prob_w || !prob_b)) return MVSTER_ERR_NULL;
    if ((inv_min == nullptr) != (inv_max == nullptr)) return MVSTER_ERR_NULL;
    if (B <= 0 || D < 1 || D > kMaxSelectD || h <= 0 || w <= 0) return MVSTER_ERR_SHAPE;
    if (inv_min && D < 3) return MVSTER_ERR_SHAPE;
    if (feat && (CF <= 0 || CF % 4 != 0)) return MVSTER_ERR_SHAPE;
    SelectArgs a;
    a.logits = logits; a.feat = feat; a.prob_w = prob_w; a.prob_b = prob_b; a.hypo = hypo; a.attn = attn;
    a.depth = depth; a.conf = conf; a.inv_min = inv_min; a.inv_max = inv_max; a.logits_out = logits_out;
    a.B = B; a.D = D; a.hw = h * w; a.CF = CF; a.split_itv = split_itv;
    if (D <= kMaxD) hipLaunchKernelGGL(select_depth_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(select_depth_any_kernel, dim3((h * w + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

extern "C" int mvster_upsample_bilinear(const float* in, float* out, int B, int hi, int wi, int ho, int wo,
                                        void* stream) {
    if (!in || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || hi <= 0 || wi <= 0 || ho <= 0 || wo <= 0) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3((ho * wo + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, in,
                       out, B, hi, wi, ho, wo);
    return mv_check_launch();
}

extern "C" int mvster_fpn_tail_gather(const float* G, const float* vb, float* P, float* workspace, int NB, int H,
                                      int W, int CO, void* stream) {
    if (!G || !vb || !P) return MVSTER_ERR_NULL;
    if (NB <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return MVSTER_ERR_SHAPE;
    if (CO != 8 && CO != 16) return MVSTER_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 block(256);
    if (workspace) {   // separable two-pass form, workspace = [NB, H, W/2, 3*CO]
        dim3 g1((H * (W / 2) + 255) / 256, NB), g2((H * W + 255) / 256, NB);
        if (CO == 8) {
            hipLaunchKernelGGL(fpn_tail_vpass_kernel<8>, g1, block, 0, s, G, workspace, NB, H, W);
            hipLaunchKernelGGL(fpn_tail_hpass_kernel<8>, g2, block, 0, s, workspace, vb, P, NB, H, W);
        } else {
            hipLaunchKernelGGL(fpn_tail_vpass_kernel<16>, g1, block, 0, s, G, workspace, NB, H, W);
            hipLaunchKernelGGL(fpn_tail_hpass_kernel<16>, g2, block, 0, s, workspace, vb, P, NB, H, W);
        }
        return mv_check_launch();
    }
    if (H >= 16 && W >= 64 && (long)(H / 2) * (W / 2) * 9 * CO * 4 < (1L << 31)) {   // LDS-tiled gather (8 x 32 output tiles, 8 channels per workgroup; 32-bit offsets)
        const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
        if (CO == 8)
            hipLaunchKernelGGL(fpn_tail_gather_lds_kernel<8>, dim3(tiles_x * tiles_y * NB), block, 0, s, G, vb, P, NB, H, W,
                               mv_fastdiv(tiles_x), mv_fastdiv(tiles_y), mv::lerp_scale(H / 2, H), mv::lerp_scale(W / 2, W));
        else
            // (probe build: MVSTER_GATHER_PAD = bytes of unused dynamic LDS per workgroup -- fewer workgroups per CU, room for the
            //  other depth map's kernels; measured neutral, profiles/r06_k_inflight_tune.txt)
            hipLaunchKernelGGL(fpn_tail_gather_lds_kernel<16>, dim3(tiles_x * tiles_y * NB, 2), block,
                               (size_t)(MV_PROBE_ENV("MVSTER_GATHER_PAD") ? atoi(MV_PROBE_ENV("MVSTER_GATHER_PAD")) : 0), s, G, vb, P, NB, H,
                               W, mv_fastdiv(tiles_x), mv_fastdiv(tiles_y), mv::lerp_scale(H / 2, H), mv::lerp_scale(W / 2, W));
        return mv_check_launch();
    }
    dim3 grid((H * W + 255) / 256, NB);
    if (CO == 8) hipLaunchKernelGGL(fpn_tail_gather_kernel<8>, grid, block, 0, s, G, vb, P, NB, H, W);
    else hipLaunchKernelGGL(fpn_tail_gather_kernel<16>, grid, block, 0, s, G, vb, P, NB, H, W);
    return mv_check_launch();
}

// gP [NB,H,W,CO] -> gG [NB,H/2,W/2,pitch]: the adjoint of mvster_fpn_tail_gather with respect to G; pitch >= 9*CO, a
// multiple of 4 (the channels beyond 9*CO are zero-filled).  CO in {8, 16}.
extern "C" int mvster_fpn_tail_gather_bwd(const float* gP, float* gG, int NB, int H, int W, int CO, int pitch, void* stream) {
    if (!gP || !gG) return MVSTER_ERR_NULL;
    if (NB <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1) || pitch < 9 * CO || (pitch & 3)) return MVSTER_ERR_SHAPE;
    if (CO != 8 && CO != 16) return MVSTER_ERR_UNSUPPORTED;
    const long total = (long)NB * (H / 2) * (W / 2) * 9;
    if (total >= (1L << 31)) return MVSTER_ERR_SHAPE;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (CO == 8 && H >= 32 && W >= 128) {
        // (maps with at least a few tiles per CU: the LDS-tiled form; the same bits)
        const int tiles_x = (W / 2 + kGbTQX - 1) / kGbTQX, tiles_y = (H / 2 + kGbTQY - 1) / kGbTQY;
        if (pitch > 80) return MVSTER_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(fpn_tail_gather_bwd_lds_kernel<8>, dim3((unsigned)(tiles_x * tiles_y * NB)), dim3(kGbThreads), 0, s, gP, gG,
                           NB, H, W, pitch, tiles_x, tiles_y);
        return mv_check_launch();
    }
    if (CO == 8) hipLaunchKernelGGL(fpn_tail_gather_bwd_kernel<8>, grid, block, 0, s, gP, gG, NB, H, W, pitch, mv_fastdiv(W / 2), mv_fastdiv(H / 2));
    else hipLaunchKernelGGL(fpn_tail_gather_bwd_kernel<16>, grid, block, 0, s, gP, gG, NB, H, W, pitch, mv_fastdiv(W / 2), mv_fastdiv(H / 2));
    return mv_check_launch();
}

// x [NB,H,W,CI], A [CO,CI], bias [CO], q [NB,H/2,W/2,CO] -> out [NB,H,W,CO]; (CI, CO) in {(16,72), (8,72)}.
extern "C" int mvster_fpn_lateral_up(const float* x, const float* A, const float* bias, const float* q, float* out, int NB,
                                     int H, int W, int CI, int CO, void* stream) {
    if (!x || !A || !bias || !q || !out) return MVSTER_ERR_NULL;
    if (NB <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return MVSTER_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    constexpr int per_block = kLateralSlots * kLateralIters;
    dim3 grid((H * W + per_block - 1) / per_block, NB), block(72 / 4 * kLateralSlots);
    if (CI == 16 && CO == 72) hipLaunchKernelGGL((fpn_lateral_up_kernel<16, 72>), grid, block, 0, s, x, A, bias, q, out, H, W, mv_fastdiv(W));
    else if (CI == 8 && CO == 72) hipLaunchKernelGGL((fpn_lateral_up_kernel<8, 72>), grid, block, 0, s, x, A, bias, q, out, H, W, mv_fastdiv(W));
    else return MVSTER_ERR_UNSUPPORTED;
    return mv_check_launch();
}

// mvster_fpn_lateral_up (16 -> 72) followed by mvster_fpn_tail_gather (CO = 8) in one launch, the 72-channel map kept in
// LDS: x [NB,H/2,W/2,16], A [72,16], bias [72], q [NB,H/4,W/4,72], vb [9,8] -> P [NB,H,W,8].  H, W multiples of 4,
// H >= 16, W >= 64 (the LDS-tiled gather's domain); MVSTER_ERR_UNSUPPORTED otherwise (the two launches cover the rest).
extern "C" int mvster_fpn_tail_fused(const float* x, const float* A, const float* bias, const float* q, const float* vb,
                                     float* P, int NB, int H, int W, int CI, void* stream) {
    if (!x || !A || !bias || !q || !vb || !P) return MVSTER_ERR_NULL;
    if (NB <= 0 || H < 4 || W < 4) return MVSTER_ERR_SHAPE;
    if (CI != 16 || (H & 3) || (W & 3) || H < 16 || W < 64) return MVSTER_ERR_UNSUPPORTED;
    // 16 x 32 output tiles (eight waves, 78 KB of LDS, two workgroups per CU): the half-resolution patch of a tile carries
    // 209 pixels for 128 interior ones instead of 133 for 64 (8 x 32 tiles, the round-4 form; probe switch MVSTER_FPN_TILE = 8 / 32)
    // (same box, alternating, profiles/r05_fpn_tile_ab.txt: 8 rows 87.8 us / 1 145.5 depth-maps/s, 16 rows 79.3 / 1 154.0;
    //  32 rows -- sixteen waves, one workgroup per CU -- 83.8 us against 77.3 and the same 1 162 depth-maps/s)
    static const int tile_env = [] { const char* e = MV_PROBE_ENV("MVSTER_FPN_TILE"); return e ? atoi(e) : 0; }();
    const int TH = tile_env == 8 || tile_env == 32 ? tile_env : 16;
    const int tiles_x = (W + 31) / 32, tiles_y = (H + TH - 1) / TH;
    if ((long)tiles_x * tiles_y * NB >= (1L << 31)) return MVSTER_ERR_SHAPE;
    const dim3 grid(tiles_x * tiles_y * NB);
    const FastDiv dx = mv_fastdiv(tiles_x), dy = mv_fastdiv(tiles_y);
    const float sy = mv::lerp_scale(H / 2, H), sx = mv::lerp_scale(W / 2, W), sqy = mv::lerp_scale(H / 4, H / 2),
                sqx = mv::lerp_scale(W / 4, W / 2);
    auto big_lds = [](const void* kern, size_t lds, unsigned long& done) {     // above the 64 KB default: once per device
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (dev >= 0 && dev < 64 && ((done >> dev) & 1ul)) return true;
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        if (dev >= 0 && dev < 64) done |= 1ul << dev;
        return true;
    };
#ifdef MVSTER_PROBES      // the measured-but-not-chosen tile shapes compile into the probe library only
    if (TH == 8) {
        constexpr size_t lds = (size_t)(7 * 19 * 19 + 6 * 11 * 10) * 16 + 9 * 8 * 4;
        MV_NOTE_KERNEL("fpn_tail_fused_kernel<16, 8>");
        hipLaunchKernelGGL((fpn_tail_fused_kernel<16, 8>), grid, dim3(256), lds, (hipStream_t)stream, x, A, bias, q, vb, P, NB, H, W,
                           dx, dy, sy, sx, sqy, sqx);
        return mv_check_launch();
    }
    if (TH == 32) {
        constexpr size_t lds = (size_t)(19 * 19 * 19 + 11 * 11 * 10) * 16 + 9 * 8 * 4;
        static unsigned long done = 0;
        if (!big_lds(reinterpret_cast<const void*>(fpn_tail_fused_kernel<16, 32>), lds, done)) return MVSTER_ERR_LAUNCH;
        MV_NOTE_KERNEL("fpn_tail_fused_kernel<16, 32>");
        hipLaunchKernelGGL((fpn_tail_fused_kernel<16, 32>), grid, dim3(1024), lds, (hipStream_t)stream, x, A, bias, q, vb, P, NB, H, W,
                           dx, dy, sy, sx, sqy, sqx);
        return mv_check_launch();
    }
#endif
    // 77 904 B: two workgroups (sixteen waves) per CU.  With 8 KB of padding (ONE workgroup per CU) the other depth map's kernels
    // find half of the LDS free and the two-in-flight rate gains 0.4 % (1 179.9 against 1 174.9 depth-maps/s over six alternating
    // runs) -- but the kernel itself takes 107 instead of 80 us: not adopted (profiles/r06_k_inflight_tune.txt).
    // (probe build: MVSTER_FUSED_PAD = bytes of such padding)
    size_t pad = 0;
    if (const char* e = MV_PROBE_ENV("MVSTER_FUSED_PAD")) pad = (size_t)atoi(e);
    const size_t lds = (size_t)(11 * 19 * 19 + 8 * 11 * 10) * 16 + 9 * 8 * 4 + pad;
    static unsigned long done = 0;
    if (!big_lds(reinterpret_cast<const void*>(fpn_tail_fused_kernel<16, 16>), lds, done)) return MVSTER_ERR_LAUNCH;
    MV_NOTE_KERNEL("fpn_tail_fused_kernel<16, 16>");
    hipLaunchKernelGGL((fpn_tail_fused_kernel<16, 16>), grid, dim3(512), lds, (hipStream_t)stream, x, A, bias, q, vb, P, NB, H, W, dx,
                       dy, sy, sx, sqy, sqx);
    return mv_check_launch();
}

extern "C" int mvster_pack_images(const float* const* imgs, int N, float* out, int B, int H, int W, void* stream) {
    if (!imgs || !out) return MVSTER_ERR_NULL;
    if (N < 1 || N > 16 || B <= 0 || H <= 0 || W <= 0) return MVSTER_ERR_SHAPE;
    PackArgs a;
    for (int v = 0; v < N; ++v) {
        if (!imgs[v]) return MVSTER_ERR_NULL;
        a.img[v] = imgs[v];
    }
    a.out = out; a.N = N; a.B = B; a.HW = H * W;
    hipLaunchKernelGGL(pack_images_kernel, dim3((H * W + 255) / 256, N * B), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

extern "C" int mvster_relative_projection_multi(const float* const* proj_matrices, int nstage, float* rt, int B, int N,
                                                void* stream) {
    if (!proj_matrices || !rt) return MVSTER_ERR_NULL;
    if (nstage < 1 || nstage > 8 || B <= 0 || N < 2) return MVSTER_ERR_SHAPE;
    MultiProjArgs a;
    for (int s = 0; s < nstage; ++s) {
        if (!proj_matrices[s]) return MVSTER_ERR_NULL;
        a.pm[s] = proj_matrices[s];
    }
    a.rt = rt; a.nstage = nstage; a.B = B; a.N = N;
    const int n = nstage * B * (N - 1);
    hipLaunchKernelGGL(relative_projection_multi_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

extern "C" int mvster_forward_prologue(const float* const* imgs, int N, float* packed, int B, int H, int W,
                                       const float* const* proj_matrices, int nstage, float* rt, const float* depth_values,
                                       int ndv, float* hypo, int D, int h, int w, int inverse, void* stream) {
    if (!imgs || !packed || !proj_matrices || !rt || !depth_values || !hypo) return MVSTER_ERR_NULL;
    if (N < 2 || N > 16 || B <= 0 || H <= 0 || W <= 0 || nstage < 1 || nstage > 8 || D < 2 || h <= 0 || w <= 0 || ndv < 1)
        return MVSTER_ERR_SHAPE;
    // (the side jobs ride on the pack grid's x extent: one row of workgroups each)
    if ((long)h * w > (long)H * W || (long)nstage * B * (N - 1) > (long)H * W) return MVSTER_ERR_SHAPE;
    PrologueArgs a;
    for (int v = 0; v < N; ++v) {
        if (!imgs[v]) return MVSTER_ERR_NULL;
        a.pack.img[v] = imgs[v];
    }
    for (int s = 0; s < nstage; ++s) {
        if (!proj_matrices[s]) return MVSTER_ERR_NULL;
        a.proj.pm[s] = proj_matrices[s];
    }
    a.pack.out = packed; a.pack.N = N; a.pack.B = B; a.pack.HW = H * W;
    a.proj.rt = rt; a.proj.nstage = nstage; a.proj.B = B; a.proj.N = N;
    a.dv = depth_values; a.hypo = hypo; a.ndv = ndv; a.D = D; a.hw = h * w; a.inverse = inverse;
    hipLaunchKernelGGL(forward_prologue_kernel, dim3((H * W + 255) / 256, N * B + 2), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

extern "C" int mvster_upsample_bilinear_multi(const float* const* ins, float* const* outs, const int* his, const int* wis,
                                              int n, int B, int ho, int wo, void* stream) {
    if (!ins || !outs || !his || !wis) return MVSTER_ERR_NULL;
    if (n < 1 || n > 8 || B <= 0 || ho <= 0 || wo <= 0) return MVSTER_ERR_SHAPE;
    MultiUpArgs a;
    for (int k = 0; k < n; ++k) {
        if (!ins[k] || !outs[k]) return MVSTER_ERR_NULL;
        if (his[k] <= 0 || wis[k] <= 0) return MVSTER_ERR_SHAPE;
        a.in[k] = ins[k]; a.out[k] = outs[k]; a.hi[k] = his[k]; a.wi[k] = wis[k];
    }
    a.ho = ho; a.wo = wo;
    hipLaunchKernelGGL(upsample_bilinear_multi_kernel, dim3((ho * wo + 255) / 256, B, n), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

// Backward of (1x1x1 `prob` head) + softmax over depth with respect to the 8-channel feature volume and the head's
// parameters (training; autograd of models/mvs4net_utils.py:900 and :1068): given d L / d attn,
//   dlogit[d] = attn[d] * (g[d] - sum_e attn[e] g[e]);  dfeat[d][c] = dlogit[d] * w[c];  dw[c] = sum dlogit feat[c];  db = sum dlogit
// One thread per pixel walks the D hypotheses; dw / db are reduced per workgroup into partial[block][CF + 1] (summed by
// the caller in a fixed order: deterministic).  Replaces ~10 tensor-level kernels per stage (two of them passes over the
// 84 MB feature volume of the full-resolution stage).
namespace {
template <int CF>
__global__ void __launch_bounds__(256) select_depth_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ gattn,
                                                               const float* __restrict__ feat, const float* __restrict__ prob_w,
                                                               float* __restrict__ dfeat, float* __restrict__ partial, int D,
                                                               long hw) {
    __shared__ float red[4][CF + 1];
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    float dw[CF + 1];
#pragma unroll
    for (int c = 0; c <= CF; ++c) dw[c] = 0.0f;
    if (p < hw) {
        const long vol = (long)b * D * hw;
        float a[mv::kSelMaxD], g[mv::kSelMaxD];
        float dot = 0.0f;
#pragma unroll
        for (int d = 0; d < mv::kSelMaxD; ++d) {
            if (d >= D) break;
            a[d] = attn[vol + d * hw + p];
            g[d] = gattn[vol + d * hw + p];
            dot = fmaf(a[d], g[d], dot);
        }
        f32x4 w4[CF / 4];
#pragma unroll
        for (int c = 0; c < CF / 4; ++c) w4[c] = ld4(prob_w + 4 * c);
#pragma unroll
        for (int d = 0; d < mv::kSelMaxD; ++d) {
            if (d >= D) break;
            const float dl = a[d] * (g[d] - dot);
            const long o = (vol + d * hw + p) * CF;
#pragma unroll
            for (int c = 0; c < CF / 4; ++c) {
                const f32x4 f = ld4(feat + o + 4 * c);
                st4(dfeat + o + 4 * c, (f32x4){dl * w4[c][0], dl * w4[c][1], dl * w4[c][2], dl * w4[c][3]});
#pragma unroll
                for (int j = 0; j < 4; ++j) dw[4 * c + j] = fmaf(dl, f[j], dw[4 * c + j]);
            }
            dw[CF] += dl;
        }
    }
#pragma unroll
    for (int c = 0; c <= CF; ++c) {
        float v = dw[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if (threadIdx.x <= CF)
        partial[((long)blockIdx.y * gridDim.x + blockIdx.x) * (CF + 1) + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
}  // namespace

// attn, gattn [B,D,h,w]; feat, dfeat [B,D,h,w,8]; prob_w [8]; partial [B * ceil(hw/256), 9] (dw[0..7], db): the caller sums
// its rows.  D <= 16.
extern "C" int mvster_select_depth_bwd(const float* attn, const float* gattn, const float* feat, const float* prob_w,
                                       float* dfeat, float* partial, int B, int D, int h, int w, int CF, void* stream) {
    if (!attn || !gattn || !feat || !prob_w || !dfeat || !partial) return MVSTER_ERR_NULL;
    if (B <= 0 || D < 1 || D > mv::kSelMaxD || h <= 0 || w <= 0) return MVSTER_ERR_SHAPE;
    if (CF != 8) return MVSTER_ERR_UNSUPPORTED;
    const long hw = (long)h * w;
    hipLaunchKernelGGL(select_depth_bwd_kernel<8>, dim3((unsigned)((hw + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, attn,
                       gattn, feat, prob_w, dfeat, partial, D, hw);
    return mv_check_launch();
}

thread_local const char* mv_last_kernel = "";

// Name (profiler spelling, template arguments included) of the kernel the most recent mvster_conv_mfma / mvster_conv_small /
// mvster_deconv_small / mvster_warp_agg_fwd call on this thread launched; "" before the first one.
extern "C" const char* mvster_last_kernel() { return mv_last_kernel; }

// bit 0: probe build (-DMVSTER_PROBES: experiment switches honoured, kernel forms kept for the record compiled in)
extern "C" int mvster_build_flags() {
#ifdef MVSTER_PROBES
    return 1;
#else
    return 0;
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// Batched gather: dst[i] = idx[i] > 0 ? src[idx[i] - 1] : 0 for a table of (src, dst, idx, n) records, ONE launch.
// The training step refreshes ~130 packed / permuted weight arrays after every optimizer update (forward and
// input-gradient forms of every layer): as one launch per array that is ~0.5 ms of 3-us kernels inside the captured step.
// Every such array is a fixed permutation (+ zero padding) of its parameter, so the host records the permutation once
// (mvster_amd/train_ops.py: the existing pack routines run on a tensor of its own indices) and a step replays this kernel.
// Replaces nothing in the reference (its layers read nn.Parameter tensors directly); host-side plumbing of the packed forms.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct GatherDesc {
    const float* src;
    float* dst;
    const int* idx;
    int n;              // elements (a multiple of 4)
    int first_block;    // prefix sum of ceil(n / 1024) over the records before this one
};

__global__ void __launch_bounds__(256) gather_batch_kernel(const GatherDesc* __restrict__ descs, int ndesc) {
    // the record this block belongs to: largest r with first_block[r] <= blockIdx.x
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const GatherDesc d = descs[lo];
    const int i = ((int)blockIdx.x - d.first_block) * 1024 + threadIdx.x * 4;
    if (i >= d.n) return;
    const int4 ix = *reinterpret_cast<const int4*>(d.idx + i);
    f32x4 v;
    v[0] = ix.x > 0 ? d.src[ix.x - 1] : 0.0f;
    v[1] = ix.y > 0 ? d.src[ix.y - 1] : 0.0f;
    v[2] = ix.z > 0 ? d.src[ix.z - 1] : 0.0f;
    v[3] = ix.w > 0 ? d.src[ix.w - 1] : 0.0f;
    st4(d.dst + i, v);
}
}  // namespace

// descs: DEVICE array of ndesc records {const float* src; float* dst; const int* idx; int n; int first_block} (32 bytes each,
// n % 4 == 0, dst and idx 16-byte aligned); total_blocks = sum of ceil(n / 1024).
extern "C" int mvster_gather_batch(const void* descs, int ndesc, int total_blocks, void* stream) {
    if (!descs) return MVSTER_ERR_NULL;
    if (ndesc <= 0 || total_blocks <= 0) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(gather_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const GatherDesc*>(descs), ndesc);
    return mv_check_launch();
}

