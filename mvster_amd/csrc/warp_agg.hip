// Fused homography warp + group correlation + epipolar attention aggregation (forward).
//
// Replaces, for one cascade stage and ALL source views in one launch, the reference's
//   homo_warping            models/mvs4net_utils.py:13-59
//   group / squared corr    models/mvs4net_utils.py:1037-1042
//   attention aggregation   models/mvs4net_utils.py:1048-1060
// without ever materialising the [B,C,D,h,w] warped volumes.
//
// Data layout (HBM):
//   ref feature  [B, h, w, C]        channels-last fp32
//   src features [NV][B, Hs, Ws, C]  channels-last fp32 (view / batch strides are arguments)
//   rt           [B, NV, 12]         relative projection rot(9) + trans(3)
//   hypo         [B, D, h, w]        depth hypotheses (API layout of `hypo_depth`)
//   out          [B, D, h, w, G]     aggregated correlation, channels-last (feeds conv0 of reg2d)
//
// Mapping: a workgroup is 64 consecutive reference pixels x D hypotheses; wave d handles
// hypothesis d for the 64 pixels (lane = pixel, so a wave's four bilinear taps fall on a
// short, contiguous run of source texels: one channels-last tap is C*4 bytes per lane and
// neighbouring lanes hit neighbouring texels).  The softmax over the depth axis is the
// only cross-wave step: each wave publishes its 64 scores in LDS (double-buffered by
// view parity, one barrier per view) and every lane reduces the D scores of its pixel.
//
// Roofline: HBM-bound.  Algorithmic bytes per launch = 4*[(NV+1)*C*hw + D*hw + G*D*hw]*B.
#include <stdlib.h>
#include <algorithm>

#include "common.hpp"

namespace {

constexpr int kMaxD = 16;
constexpr int kMaxFwdD = 64;      // forward only (warp_agg_fwd_kernel with fewer pixels per workgroup)

struct WarpAggArgs {
    const float* ref;
    const float* src;
    const float* rt;
    const float* hypo;
    float* out;
    float* wsum_out;  // optional [B, D, h, w] (saved for the backward pass)
    long ref_bs;      // batch stride of ref (elements)
    long src_vs;      // view stride of src
    long src_bs;      // batch stride of src
    int B, NV, D, h, w, Hs, Ws;
    float attn_temp;
    float sqrt_c;
    int fuse_d;
    // fused hypothesis scheduling (wave-local kernel, SCHED != 0): the hypotheses are computed here instead of read, and
    // written to hypo_out [B, D, h, w] for the stage's selection and the API
    const float* inv_min;    // SCHED 1: previous stage's inverse_min_depth [B, h/2, w/2]
    const float* inv_max;    //          ... inverse_max_depth
    const float* dvals;      // SCHED 2: depth_values [B, ndv] (first and last column = the range)
    float* hypo_out;
    int ndv;
};

// PX pixels x D hypotheses per workgroup: 64 pixels for D <= 16 (the shipped cascade's fallback form), 32 / 16 pixels for up
// to 32 / 64 hypotheses per stage (free --ndepths of the reference; evaluation only, the backward keeps D <= 16).
template <int C, int G, bool GROUP, int DMAX, int PX = 64>
__global__ void __launch_bounds__(PX * DMAX) warp_agg_fwd_kernel(WarpAggArgs a) {
    static_assert(C % 8 == 0, "channels-last taps are read as float4 pairs");
    static_assert(GROUP ? (C % G == 0) : (C == G), "group layout");
    constexpr int CG = C / G;                // channels per correlation group
    constexpr int CB = CG > 8 ? CG : 8;      // channels gathered per loop iteration
    constexpr int GB = CB / CG;              // groups completed per iteration
    static_assert(C % CB == 0, "channel block");
    // per-view correlations live in LDS (G floats per thread) so that the gather loop can
    // stay rolled: ~45 VGPRs for every C instead of 4*C registers of in-flight taps
    __shared__ float sc[2][DMAX][PX];
    __shared__ float corL[G][DMAX * PX];

    const int tx = threadIdx.x;
    const int d = threadIdx.y;
    const int tid = d * PX + tx;
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const int p = xcd_remap(blockIdx.x, gridDim.x) * PX + tx;
    const bool valid = p < hw;
    const int pc = valid ? p : hw - 1;  // clamped: every lane takes part in the barriers
    const int y = pc / a.w;
    const int x = pc - y * a.w;
    const float depth = a.hypo[((long)b * a.D + d) * hw + pc];
    const float* rp = a.ref + (long)b * a.ref_bs + (long)pc * C;

    float acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.0f;
    float wsum = 1e-8f;

    for (int v = 0; v < a.NV; ++v) {
        mv::RT m;
        {
            const float* r = a.rt + ((long)b * a.NV + v) * 12;  // wave-uniform
#pragma unroll
            for (int i = 0; i < 9; ++i) m.r[i] = r[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) m.t[i] = r[9 + i];
        }
        float sx, sy;
        mv::project(m, (float)x, (float)y, depth, a.Hs, a.Ws, sx, sy);
        mv::Taps t = mv::make_taps(sx, sy, a.Hs, a.Ws);
        const mv::TapsClamped tc = mv::clamp_taps(t, a.Hs, a.Ws);
        const float* sp = a.src + (long)v * a.src_vs + (long)b * a.src_bs;
        const float* p00 = sp + ((long)tc.ya * a.Ws + tc.xa) * C;
        const float* p01 = sp + ((long)tc.ya * a.Ws + tc.xb) * C;
        const float* p10 = sp + ((long)tc.yb * a.Ws + tc.xa) * C;
        const float* p11 = sp + ((long)tc.yb * a.Ws + tc.xb) * C;

        float score = 0.0f;
#pragma unroll 2
        for (int cb = 0; cb < C / CB; ++cb) {
            const int cbase = cb * CB;
            float part[GB];
#pragma unroll
            for (int c0 = 0; c0 < CB; c0 += 4) {
                const f32x4 R = ld4(rp + cbase + c0);
                const f32x4 A = ld4(p00 + cbase + c0);
                const f32x4 Bq = ld4(p01 + cbase + c0);
                const f32x4 Cq = ld4(p10 + cbase + c0);
                const f32x4 Dq = ld4(p11 + cbase + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + j;  // channel within the block
                    const float wv = mv::blend(t, A[j], Bq[j], Cq[j], Dq[j]);
                    if (GROUP) {
                        const float pr = mv::mul_rn(wv, R[j]);
                        part[c / CG] = (c % CG == 0) ? pr : mv::add_rn(part[c / CG], pr);
                    } else {
                        const float df = mv::sub_rn(R[j], wv);
                        part[c] = mv::mul_rn(df, df);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const float cg = GROUP ? mv::div_rn(part[k], (float)CG) : part[k];  // .mean(2)
                corL[cb * GB + k][tid] = cg;
                score = (cb == 0 && k == 0) ? cg : mv::add_rn(score, cg);           // .sum(1)
            }
        }
        if (a.fuse_d) score = mv::div_rn(score, a.attn_temp);

        float (*buf)[PX] = sc[v & 1];
        buf[d][tx] = score;
        __syncthreads();
        float mx = buf[0][tx];
        for (int j = 1; j < a.D; ++j) mx = fmaxf(mx, buf[j][tx]);
        float den = 0.0f;
        for (int j = 0; j < a.D; ++j) den = mv::add_rn(den, expf(mv::sub_rn(buf[j][tx], mx)));
        float wgt;
        if (a.fuse_d)
            wgt = mv::div_rn(mv::div_rn(expf(mv::sub_rn(score, mx)), den), a.sqrt_c);
        else
            wgt = mv::div_rn(1.0f, den);  // max_d softmax = exp(0) / sum
        wsum = mv::add_rn(wsum, wgt);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = mv::add_rn(acc[g], mv::mul_rn(wgt, corL[g][tid]));
    }

    if (valid) {
        const long o = (((long)b * a.D + d) * hw + p);
        float* op = a.out + o * G;
#pragma unroll
        for (int g = 0; g < G; g += 4) {
            f32x4 r;
            r[0] = mv::div_rn(acc[g], wsum);
            r[1] = mv::div_rn(acc[g + 1], wsum);
            r[2] = mv::div_rn(acc[g + 2], wsum);
            r[3] = mv::div_rn(acc[g + 3], wsum);
            st4(op + g, r);
        }
        if (a.wsum_out) a.wsum_out[o] = wsum;
    }
}

// ------------------------------------------------------------------------------------------
// Lane-split variant for wide feature maps (C >= 16, group correlation).  At the coarse stages the
// map is small (5120 pixels at stage 1) but C is 64: one thread per (pixel, d) leaves most CUs idle
// and makes each thread walk 8 dependent gather rounds.  Here LPP = C/8 adjacent lanes share one
// (pixel, d): lane `sub` gathers channels 8*sub..8*sub+7 (= whole correlation groups, since C/G <= 8),
// so a tap of one pixel is LPP*32 contiguous bytes across neighbouring lanes, the chip sees LPP x more
// waves, and there is a single gather round per view.  The G correlations are all-gathered with wave
// shuffles and summed in group order, so scores are bit-identical to the one-thread form.
// ------------------------------------------------------------------------------------------
template <int C, int G, int DMAX>
__global__ void __launch_bounds__(64 * DMAX) warp_agg_fwd_lanes_kernel(WarpAggArgs a) {
    constexpr int LPP = C / 8;           // lanes per (pixel, d)
    constexpr int CG = C / G;            // channels per group
    constexpr int GPL = 8 / CG;          // whole groups per lane
    constexpr int PPB = 64 / LPP;        // pixels per workgroup
    static_assert(C % 8 == 0 && CG <= 8 && 8 % CG == 0 && LPP >= 1 && LPP <= 16, "lane split");
    __shared__ float sc[2][DMAX][PPB];

    const int tx = threadIdx.x;
    const int sub = tx % LPP, pl = tx / LPP;
    const int d = threadIdx.y;
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const int p = xcd_remap(blockIdx.x, gridDim.x) * PPB + pl;
    const bool valid = p < hw;
    const int pc = valid ? p : hw - 1;
    const int y = pc / a.w;
    const int x = pc - y * a.w;
    const float depth = a.hypo[((long)b * a.D + d) * hw + pc];
    const float* rp = a.ref + (long)b * a.ref_bs + (long)pc * C + sub * 8;
    const f32x4 R0 = ld4(rp), R1 = ld4(rp + 4);

    float acc[GPL];
#pragma unroll
    for (int k = 0; k < GPL; ++k) acc[k] = 0.0f;
    float wsum = 1e-8f;

    for (int v = 0; v < a.NV; ++v) {
        mv::RT m;
        {
            const float* r = a.rt + ((long)b * a.NV + v) * 12;
#pragma unroll
            for (int i = 0; i < 9; ++i) m.r[i] = r[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) m.t[i] = r[9 + i];
        }
        float sx, sy;
        mv::project(m, (float)x, (float)y, depth, a.Hs, a.Ws, sx, sy);
        mv::Taps t = mv::make_taps(sx, sy, a.Hs, a.Ws);
        const mv::TapsClamped tc = mv::clamp_taps(t, a.Hs, a.Ws);
        const float* sp = a.src + (long)v * a.src_vs + (long)b * a.src_bs + sub * 8;
        const float* p00 = sp + ((long)tc.ya * a.Ws + tc.xa) * C;
        const float* p01 = sp + ((long)tc.ya * a.Ws + tc.xb) * C;
        const float* p10 = sp + ((long)tc.yb * a.Ws + tc.xa) * C;
        const float* p11 = sp + ((long)tc.yb * a.Ws + tc.xb) * C;
        const f32x4 A0 = ld4(p00), A1 = ld4(p00 + 4), B0 = ld4(p01), B1 = ld4(p01 + 4);
        const f32x4 C0 = ld4(p10), C1 = ld4(p10 + 4), D0 = ld4(p11), D1 = ld4(p11 + 4);

        float part[GPL];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float wv = c < 4 ? mv::blend(t, A0[c & 3], B0[c & 3], C0[c & 3], D0[c & 3])
                                   : mv::blend(t, A1[c & 3], B1[c & 3], C1[c & 3], D1[c & 3]);
            const float pr = mv::mul_rn(wv, c < 4 ? R0[c & 3] : R1[c & 3]);
            part[c / CG] = (c % CG == 0) ? pr : mv::add_rn(part[c / CG], pr);
        }
        float cg[GPL];
#pragma unroll
        for (int k = 0; k < GPL; ++k) cg[k] = mv::div_rn(part[k], (float)CG);

        // all-gather the G correlations of this (pixel, d) and sum them in group order
        float score = 0.0f;
        const int lane0 = tx - sub;
#pragma unroll
        for (int j = 0; j < LPP; ++j)
#pragma unroll
            for (int k = 0; k < GPL; ++k) {
                const float val = __shfl(cg[k], lane0 + j);
                score = (j == 0 && k == 0) ? val : mv::add_rn(score, val);
            }
        if (a.fuse_d) score = mv::div_rn(score, a.attn_temp);

        float (*buf)[PPB] = sc[v & 1];
        if (sub == 0) buf[d][pl] = score;
        __syncthreads();
        float mx = buf[0][pl];
        for (int j = 1; j < a.D; ++j) mx = fmaxf(mx, buf[j][pl]);
        float den = 0.0f;
        for (int j = 0; j < a.D; ++j) den = mv::add_rn(den, expf(mv::sub_rn(buf[j][pl], mx)));
        float wgt;
        if (a.fuse_d)
            wgt = mv::div_rn(mv::div_rn(expf(mv::sub_rn(score, mx)), den), a.sqrt_c);
        else
            wgt = mv::div_rn(1.0f, den);
        wsum = mv::add_rn(wsum, wgt);
#pragma unroll
        for (int k = 0; k < GPL; ++k) acc[k] = mv::add_rn(acc[k], mv::mul_rn(wgt, cg[k]));
    }

    if (valid) {
        const long o = (((long)b * a.D + d) * hw + p);
        float* op = a.out + o * G + sub * GPL;
#pragma unroll
        for (int k = 0; k < GPL; ++k) op[k] = mv::div_rn(acc[k], wsum);
        if (a.wsum_out && sub == 0) a.wsum_out[o] = wsum;
    }
}

// ------------------------------------------------------------------------------------------
// Wave-local variant (the default whenever D and C/8 are powers of two with D*C/8 <= 64, i.e. all
// four stages of the shipped cascade).  One wavefront owns PPW = 64/(D*C/8) pixels and ALL their
// depth hypotheses: lane = (d*PPW + pixel)*LPP + sub.  The softmax over depth is an in-wave
// all-gather (ds_bpermute), so there is no LDS, no barrier and no workgroup-level coupling: waves
// run free and the only thing a wave ever waits for is its own gather (8 independent 16-byte loads
// per lane and view; gathering two views per iteration was measured slower, it costs occupancy).
// The kernel is VALU-bound, not HBM-bound (~200 instructions per (pixel, d, view) against 256 bytes
// gathered), hence the shared-reciprocal divisions, the single expf per lane and the packed blend.
// Arithmetic and summation order are those of the one-thread form: results are bit-identical to it.
// ------------------------------------------------------------------------------------------
// SCHED: 0 = the hypotheses are read from a.hypo; 1 = schedule_inverse_range of the previous stage's inverse bounds and
// 2 = init_inverse_range of depth_values, both computed per lane with the scheduler kernels' own per-hypothesis functions
// (mvster_math.h: bit-identical) and written to a.hypo_out by lane sub 0 -- one launch and one dependency edge less per
// stage (schedule_inverse_kernel ran alone for ~5 us three times per forward).
template <int C, int G, int D, int SCHED = 0>
__global__ void __launch_bounds__(256) warp_agg_fwd_wave_kernel(WarpAggArgs a) {
    constexpr int LPP = C / 8;           // lanes per (pixel, d)
    constexpr int CG = C / G;            // channels per group
    constexpr int GPL = 8 / CG;          // whole groups per lane
    constexpr int PPW = 64 / (LPP * D);  // pixels per wave
    static_assert(C % 8 == 0 && CG <= 8 && 8 % CG == 0 && PPW >= 1 && PPW * LPP * D == 64, "wave split");

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPP, pl = (lane / LPP) % PPW, d = lane / (LPP * PPW);
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const int p = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * PPW + pl;
    const bool valid = p < hw;
    const int pc = valid ? p : hw - 1;   // clamped: every lane takes part in the shuffles
    const int y = pc / a.w;
    const int x = pc - y * a.w;
    float depth;
    if constexpr (SCHED == 0) {
        depth = a.hypo[((long)b * D + d) * hw + pc];
    } else {
        if constexpr (SCHED == 1) {
            const int hi = a.h / 2, wi = a.w / 2;
            const mv::InvCorners cn = mv::schedule_inverse_corners(a.inv_min + (long)b * hi * wi, a.inv_max + (long)b * hi * wi,
                                                                   y, x, a.h, a.w, hi, wi);
            depth = mv::schedule_inverse_one(cn, D, d);
        } else {
            depth = mv::init_inverse_one(a.dvals[(long)b * a.ndv], a.dvals[(long)b * a.ndv + a.ndv - 1], D, d);
        }
        if (valid && sub == 0) a.hypo_out[((long)b * D + d) * hw + pc] = depth;
    }
    const float* rp = a.ref + (long)b * a.ref_bs + (long)pc * C + sub * 8;
    const f32x4 R0 = ld4(rp), R1 = ld4(rp + 4);
    // per-launch divisors with their reciprocals (mvster_math.h: same bits as '/', 5 instead of 11 VALU ops)
    const mv::GridNorm gn = mv::make_grid_norm(a.Hs, a.Ws);
    const mv::Recip temp = mv::make_recip(a.attn_temp), sqrt_c = mv::make_recip(a.sqrt_c);
    constexpr int SH = C == 8 ? 5 : (C == 16 ? 6 : (C == 32 ? 7 : 8));   // log2(bytes per texel)
    const float xhi = (float)(a.Ws + 4), yhi = (float)(a.Hs + 4);
    const unsigned src_bytes = (unsigned)a.Hs * (unsigned)a.Ws * (unsigned)(C * 4);
    const int row_bytes = a.Ws << SH;

    float acc[GPL];
#pragma unroll
    for (int k = 0; k < GPL; ++k) acc[k] = 0.0f;
    float wsum = 1e-8f;

    for (int v = 0; v < a.NV; ++v) {
        mv::RT m;
        const float* r = a.rt + ((long)b * a.NV + v) * 12;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 9; ++i) m.r[i] = r[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) m.t[i] = r[9 + i];
        float sx, sy;
        mv::project(m, (float)x, (float)y, depth, gn, sx, sy);
        // make_taps() with a single clamp instruction per coordinate (v_med3_f32; a NaN position ends up at -4)
        mv::Taps t;
        {
            const float cx = __builtin_amdgcn_fmed3f(sx, -4.0f, xhi), cy = __builtin_amdgcn_fmed3f(sy, -4.0f, yhi);
            const float fx = floorf(cx), fy = floorf(cy);
            t.x0 = (int)fx;
            t.y0 = (int)fy;
            const float wx1 = mv::sub_rn(cx, fx), wy1 = mv::sub_rn(cy, fy);
            const float wx0 = mv::sub_rn(1.0f, wx1), wy0 = mv::sub_rn(1.0f, wy1);
            const bool vx0 = (unsigned)t.x0 < (unsigned)a.Ws, vx1 = (unsigned)(t.x0 + 1) < (unsigned)a.Ws;
            const bool vy0 = (unsigned)t.y0 < (unsigned)a.Hs, vy1 = (unsigned)(t.y0 + 1) < (unsigned)a.Hs;
            t.nw = (vy0 && vx0) ? mv::mul_rn(wy0, wx0) : 0.0f;
            t.ne = (vy0 && vx1) ? mv::mul_rn(wy0, wx1) : 0.0f;
            t.sw = (vy1 && vx0) ? mv::mul_rn(wy1, wx0) : 0.0f;
            t.se = (vy1 && vx1) ? mv::mul_rn(wy1, wx1) : 0.0f;
        }
        // raw buffer loads: one 32-bit byte offset per tap row (24-bit multiply-add on the texel index), the other
        // addresses are immediate offsets; nothing is clamped -- a tap outside the map either falls outside the
        // descriptor (the hardware returns 0) or reads some other texel, and its weight is 0 in both cases
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.src + (long)v * a.src_vs + (long)b * a.src_bs), (short)0, (int)src_bytes, 0x00020000);
        const unsigned oa = (((unsigned)__mul24(t.y0, a.Ws) + (unsigned)t.x0) << SH) + (unsigned)(sub * 32);
        const unsigned ob = oa + (unsigned)row_bytes;
        const f32x4 q0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa, 0, 0));
        const f32x4 q1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + 16, 0, 0));
        const f32x4 q2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + (4 * C), 0, 0));
        const f32x4 q3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + (4 * C + 16), 0, 0));
        const f32x4 q4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob, 0, 0));
        const f32x4 q5 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + 16, 0, 0));
        const f32x4 q6 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + (4 * C), 0, 0));
        const f32x4 q7 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + (4 * C + 16), 0, 0));
        // keep the eight loads together and ahead of their uses: left alone, the scheduler sinks each
        // load to its first use to save registers and the wave then eats eight memory latencies in a row
        __builtin_amdgcn_sched_barrier(0);

        // blend and correlate channel pairs (packed fp32 on the pairs the 16-byte loads deliver), then sum
        // each group in channel order
        float part[GPL];
#define MV_PAIR(c, A, B, Cq, Dq, R, j)                                                                         \
        {                                                                                                      \
            const mv::f32x2 wv = mv::blend2(t.nw, t.ne, t.sw, t.se, (mv::f32x2){A[j], A[j + 1]},               \
                                            (mv::f32x2){B[j], B[j + 1]}, (mv::f32x2){Cq[j], Cq[j + 1]},        \
                                            (mv::f32x2){Dq[j], Dq[j + 1]});                                    \
            const mv::f32x2 p2 = mv::mul_rn2(wv, (mv::f32x2){R[j], R[j + 1]});                                 \
            part[(c) / CG] = ((c) % CG == 0) ? p2[0] : mv::add_rn(part[(c) / CG], p2[0]);                      \
            part[((c) + 1) / CG] = (((c) + 1) % CG == 0) ? p2[1] : mv::add_rn(part[((c) + 1) / CG], p2[1]);    \
        }
        MV_PAIR(0, q0, q2, q4, q6, R0, 0)
        MV_PAIR(2, q0, q2, q4, q6, R0, 2)
        MV_PAIR(4, q1, q3, q5, q7, R1, 0)
        MV_PAIR(6, q1, q3, q5, q7, R1, 2)
#undef MV_PAIR
        float cg[GPL];
#pragma unroll
        for (int k = 0; k < GPL; ++k) cg[k] = mv::div_rn(part[k], (float)CG);   // .mean(2)
        // all-gather the G correlations of this (pixel, d) and sum them in group order: .sum(1)
        float score = 0.0f;
        const int lane0 = lane - sub;
#pragma unroll
        for (int j = 0; j < LPP; ++j)
#pragma unroll
            for (int k = 0; k < GPL; ++k) {
                const float val = LPP == 1 ? cg[k] : __shfl(cg[k], lane0 + j);
                score = (j == 0 && k == 0) ? val : mv::add_rn(score, val);
            }
        if (a.fuse_d) score = mv::div_rn(score, temp);
        // softmax over depth: the D scores of this pixel live in lanes (j*PPW + pl)*LPP + sub.  Each lane
        // exponentiates its own score once and the D terms are all-gathered and summed in depth order
        // (the other launch forms evaluate the same D expf calls in every thread).
        float mx = score;
#pragma unroll
        for (int j = 0; j < D; ++j) mx = fmaxf(mx, __shfl(score, (j * PPW + pl) * LPP + sub));
        const float e = expf(mv::sub_rn(score, mx));
        float den = 0.0f;
#pragma unroll
        for (int j = 0; j < D; ++j) den = mv::add_rn(den, __shfl(e, (j * PPW + pl) * LPP + sub));
        const mv::Recip rden = mv::make_recip(den);
        float wgt;
        if (a.fuse_d)
            wgt = mv::div_rn(mv::div_rn(e, rden), sqrt_c);
        else
            wgt = mv::div_rn(1.0f, rden);  // max_d softmax = exp(0) / sum
        wsum = mv::add_rn(wsum, wgt);
#pragma unroll
        for (int k = 0; k < GPL; ++k) acc[k] = mv::add_rn(acc[k], mv::mul_rn(wgt, cg[k]));
    }

    if (valid) {
        const long o = (((long)b * D + d) * hw + p);
        float* op = a.out + o * G + sub * GPL;
        const mv::Recip rw = mv::make_recip(wsum);
        if (GPL == 4) {
            st4(op, (f32x4){mv::div_rn(acc[0], rw), mv::div_rn(acc[GPL > 1 ? 1 : 0], rw),
                            mv::div_rn(acc[GPL > 2 ? 2 : 0], rw), mv::div_rn(acc[GPL > 3 ? 3 : 0], rw)});
        } else {
#pragma unroll
            for (int k = 0; k < GPL; ++k) op[k] = mv::div_rn(acc[k], rw);
        }
        if (a.wsum_out && sub == 0) a.wsum_out[o] = wsum;
    }
}

#ifdef MVSTER_PROBES   // pixel-major form (variant 4): probe build only
// ------------------------------------------------------------------------------------------
// Pixel-major variant for the two fine stages (C <= 16, where the time is): lane = (pixel, sub), and the lane
// walks ALL D hypotheses of its pixel.  The wave-local kernel above is VALU-issue-bound (PMC: ~205 VALU
// instructions per (pixel, d, view) against 256 gathered bytes; 77 % of the SIMD issue cycles busy), so this one
// removes instructions rather than bytes:
//   * R*(x, y, 1) once per (pixel, view) instead of once per (pixel, d, view);
//   * the per-hypothesis scalar chain (r*depth + t, the IEEE division by z, the normalise / un-normalise round trip,
//     the fractional weights) runs on PAIRS of hypotheses in packed fp32 (v_pk_mul/add/fma_f32): same rounding per
//     element, half the issue slots;
//   * taps are fetched with raw buffer loads: ONE 32-bit offset per tap row ((y0*Ws + x0) << log2(4C), 24-bit
//     multiply-add), the other seven addresses are immediate offsets, and no index clamping at all -- a tap outside
//     the map either falls outside the descriptor (the hardware returns 0) or reads some other texel, and in both
//     cases its weight is already 0;
//   * the softmax over depth is in registers: no ds_bpermute, one reciprocal per pixel.
// Arithmetic per element and every summation order are those of warp_agg_fwd_kernel: bit-identical output.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ mv::f32x2 fma2(mv::f32x2 a, mv::f32x2 b, mv::f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ mv::f32x2 splat2(float v) { return (mv::f32x2){v, v}; }

struct Recip2 {
    mv::f32x2 d, r;
};
__device__ __forceinline__ Recip2 make_recip2(mv::f32x2 d) {
    Recip2 k;
    k.d = d;
    const mv::f32x2 r0 = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    k.r = fma2(fma2(-d, r0, splat2(1.0f)), r0, r0);
    return k;
}
// same sequence as mv::div_rn(float, Recip) on both elements
__device__ __forceinline__ mv::f32x2 div_rn2(mv::f32x2 a, const Recip2& k) {
    mv::f32x2 q = mv::mul_rn2(a, k.r);
    q = fma2(fma2(-k.d, q, a), k.r, q);
    return fma2(fma2(-k.d, q, a), k.r, q);
}
__device__ __forceinline__ mv::f32x2 div_rn2(mv::f32x2 a, const mv::Recip& k) {
    const mv::f32x2 d = splat2(k.d), r = splat2(k.r);
    mv::f32x2 q = mv::mul_rn2(a, r);
    q = fma2(fma2(-d, q, a), r, q);
    return fma2(fma2(-d, q, a), r, q);
}
__device__ __forceinline__ mv::f32x2 sub_rn2(mv::f32x2 a, mv::f32x2 b) {
#pragma clang fp contract(off)
    return a - b;
}

// DPL = hypotheses per lane (even): DPL = D is the form described above; DPL = 2 spreads the D / 2 hypothesis pairs of a
// pixel over QL = D / 2 lanes (twice / four times the waves, half / a quarter of the registers' worth of in-flight
// taps per lane), with the softmax over depth as an in-wave all-gather like the wave-local kernel.
template <int C, int G, int D, int DPL, int WPE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, 8))) warp_agg_fwd_pix_kernel(WarpAggArgs a) {
    constexpr int LPP = C / 8;           // lanes per (pixel, hypothesis group): channel slices of 8
    constexpr int CG = C / G;            // channels per group
    constexpr int GPL = 8 / CG;          // whole groups per lane
    constexpr int QL = D / DPL;          // lanes (hypothesis groups) per pixel and channel slice
    constexpr int PPW = 64 / (LPP * QL); // pixels per wave (a workgroup is one wave: waves never talk to each other)
    constexpr int SH = C == 8 ? 5 : (C == 16 ? 6 : (C == 32 ? 7 : 8));   // log2(bytes per texel)
    static_assert(C % 8 == 0 && CG <= 8 && 8 % CG == 0 && DPL % 2 == 0 && D % DPL == 0 && PPW >= 1 &&
                  PPW * LPP * QL == 64 && D <= 8, "pixel split");

    const int lane = threadIdx.x;
    const int sub = lane % LPP, pl = (lane / LPP) % PPW, qg = lane / (LPP * PPW);
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const int p = xcd_remap(blockIdx.x, gridDim.x) * PPW + pl;
    const bool valid = p < hw;
    const int pc = valid ? p : hw - 1;   // clamped: every lane takes part in the cross-lane sums
    const int y = pc / a.w;
    const int x = pc - y * a.w;
    const float xf = (float)x, yf = (float)y;
    const float* hp = a.hypo + ((long)b * D + qg * DPL) * hw + pc;
    mv::f32x2 depth2[DPL / 2];
#pragma unroll
    for (int q = 0; q < DPL / 2; ++q) depth2[q] = (mv::f32x2){hp[(long)(2 * q) * hw], hp[(long)(2 * q + 1) * hw]};
    const float* rp = a.ref + (long)b * a.ref_bs + (long)pc * C + sub * 8;
    const f32x4 R0 = ld4(rp), R1 = ld4(rp + 4);
    const mv::GridNorm gn = mv::make_grid_norm(a.Hs, a.Ws);
    const mv::Recip temp = mv::make_recip(a.attn_temp), sqrt_c = mv::make_recip(a.sqrt_c);
    const float xhi = (float)(a.Ws + 4), yhi = (float)(a.Hs + 4);
    const unsigned src_bytes = (unsigned)a.Hs * (unsigned)a.Ws * (unsigned)(C * 4);
    const int row_bytes = a.Ws << SH;

    float acc[DPL][GPL], wsum[DPL];
#pragma unroll
    for (int d = 0; d < DPL; ++d) {
        wsum[d] = 1e-8f;
#pragma unroll
        for (int k = 0; k < GPL; ++k) acc[d][k] = 0.0f;
    }

    for (int v = 0; v < a.NV; ++v) {
        const float* r = a.rt + ((long)b * a.NV + v) * 12;  // wave-uniform
        // rot @ (x, y, 1): the reference's sgemm FMA chain (mv::project), once per (pixel, view)
        const float rx = mv::add_rn(fmaf(r[1], yf, mv::mul_rn(r[0], xf)), r[2]);
        const float ry = mv::add_rn(fmaf(r[4], yf, mv::mul_rn(r[3], xf)), r[5]);
        const float rz = mv::add_rn(fmaf(r[7], yf, mv::mul_rn(r[6], xf)), r[8]);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.src + (long)v * a.src_vs + (long)b * a.src_bs), (short)0, (int)src_bytes, 0x00020000);

        float cg[DPL][GPL], score[DPL];
#pragma unroll
        for (int q = 0; q < DPL / 2; ++q) {
            // two hypotheses at a time through the scalar chain, packed
            const mv::f32x2 dz = depth2[q];
            const mv::f32x2 px = mv::add_rn2(mv::mul_rn2(splat2(rx), dz), splat2(r[9]));
            const mv::f32x2 py = mv::add_rn2(mv::mul_rn2(splat2(ry), dz), splat2(r[10]));
            mv::f32x2 pz = mv::add_rn2(mv::mul_rn2(splat2(rz), dz), splat2(r[11]));
            if (pz[0] == 0.0f) pz[0] = 1e-9f;
            if (pz[1] == 0.0f) pz[1] = 1e-9f;
            const Recip2 z = make_recip2(pz);
            // grid_roundtrip(): / half, - 1, + 1, * 0.5, * (size - 1)
            mv::f32x2 sx = sub_rn2(div_rn2(div_rn2(px, z), gn.halfw), splat2(1.0f));
            mv::f32x2 sy = sub_rn2(div_rn2(div_rn2(py, z), gn.halfh), splat2(1.0f));
            sx = mv::mul_rn2(mv::mul_rn2(mv::add_rn2(sx, splat2(1.0f)), splat2(0.5f)), splat2(gn.wm1));
            sy = mv::mul_rn2(mv::mul_rn2(mv::add_rn2(sy, splat2(1.0f)), splat2(0.5f)), splat2(gn.hm1));
            // make_taps(): clamp far-away / NaN positions, corner, fractional weights
            const mv::f32x2 cx = {__builtin_amdgcn_fmed3f(sx[0], -4.0f, xhi), __builtin_amdgcn_fmed3f(sx[1], -4.0f, xhi)};
            const mv::f32x2 cy = {__builtin_amdgcn_fmed3f(sy[0], -4.0f, yhi), __builtin_amdgcn_fmed3f(sy[1], -4.0f, yhi)};
            const mv::f32x2 fx = {floorf(cx[0]), floorf(cx[1])}, fy = {floorf(cy[0]), floorf(cy[1])};
            const mv::f32x2 wx1 = sub_rn2(cx, fx), wy1 = sub_rn2(cy, fy);
            const mv::f32x2 wx0 = sub_rn2(splat2(1.0f), wx1), wy0 = sub_rn2(splat2(1.0f), wy1);
            mv::f32x2 nw = mv::mul_rn2(wy0, wx0), ne = mv::mul_rn2(wy0, wx1);
            mv::f32x2 sw = mv::mul_rn2(wy1, wx0), se = mv::mul_rn2(wy1, wx1);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int d = 2 * q + e;
                const int x0 = (int)fx[e], y0 = (int)fy[e];
                const bool vx0 = (unsigned)x0 < (unsigned)a.Ws, vx1 = (unsigned)(x0 + 1) < (unsigned)a.Ws;
                const bool vy0 = (unsigned)y0 < (unsigned)a.Hs, vy1 = (unsigned)(y0 + 1) < (unsigned)a.Hs;
                const float wnw = (vy0 && vx0) ? nw[e] : 0.0f, wne = (vy0 && vx1) ? ne[e] : 0.0f;
                const float wsw = (vy1 && vx0) ? sw[e] : 0.0f, wse = (vy1 && vx1) ? se[e] : 0.0f;
                // byte offset of tap (y0, x0); |y0 * Ws + x0| < 2^24 (checked by the launcher), the shift may wrap for
                // negative corners: then the offset is out of the descriptor's range and the load returns 0
                const unsigned o0 = ((unsigned)__mul24(y0, a.Ws) + (unsigned)x0) << SH;
                const unsigned oa = o0 + (unsigned)(sub * 32), ob = oa + (unsigned)row_bytes;
                const f32x4 q0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa, 0, 0));
                const f32x4 q1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + 16, 0, 0));
                const f32x4 q2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + (4 * C), 0, 0));
                const f32x4 q3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + (4 * C + 16), 0, 0));
                const f32x4 q4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob, 0, 0));
                const f32x4 q5 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + 16, 0, 0));
                const f32x4 q6 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + (4 * C), 0, 0));
                const f32x4 q7 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + (4 * C + 16), 0, 0));
                float part[GPL];
#define MV_PAIR(c, A, B, Cq, Dq, R, j)                                                                         \
                {                                                                                              \
                    const mv::f32x2 wv = mv::blend2(wnw, wne, wsw, wse, (mv::f32x2){A[j], A[j + 1]},           \
                                                    (mv::f32x2){B[j], B[j + 1]}, (mv::f32x2){Cq[j], Cq[j + 1]}, \
                                                    (mv::f32x2){Dq[j], Dq[j + 1]});                            \
                    const mv::f32x2 p2 = mv::mul_rn2(wv, (mv::f32x2){R[j], R[j + 1]});                         \
                    part[(c) / CG] = ((c) % CG == 0) ? p2[0] : mv::add_rn(part[(c) / CG], p2[0]);              \
                    part[((c) + 1) / CG] = (((c) + 1) % CG == 0) ? p2[1] : mv::add_rn(part[((c) + 1) / CG], p2[1]); \
                }
                MV_PAIR(0, q0, q2, q4, q6, R0, 0)
                MV_PAIR(2, q0, q2, q4, q6, R0, 2)
                MV_PAIR(4, q1, q3, q5, q7, R1, 0)
                MV_PAIR(6, q1, q3, q5, q7, R1, 2)
#undef MV_PAIR
#pragma unroll
                for (int k = 0; k < GPL; ++k) cg[d][k] = mv::div_rn(part[k], (float)CG);   // .mean(2)
                // .sum(1): all groups of the (pixel, d) in group order (its LPP lanes are neighbours)
                float sc = 0.0f;
                const int lane0 = lane - sub;
#pragma unroll
                for (int j = 0; j < LPP; ++j)
#pragma unroll
                    for (int k = 0; k < GPL; ++k) {
                        const float val = LPP == 1 ? cg[d][k] : __shfl(cg[d][k], lane0 + j);
                        sc = (j == 0 && k == 0) ? val : mv::add_rn(sc, val);
                    }
                score[d] = a.fuse_d ? mv::div_rn(sc, temp) : sc;
            }
        }
        // softmax over depth (depth order, like the other launch forms): in registers, plus an in-wave all-gather over the
        // QL lanes that share the pixel when DPL < D
        float mx = score[0];
#pragma unroll
        for (int d = 1; d < DPL; ++d) mx = fmaxf(mx, score[d]);
        if (QL > 1) {
            const float own = mx;
#pragma unroll
            for (int j = 0; j < QL; ++j) mx = fmaxf(mx, __shfl(own, (j * PPW + pl) * LPP + sub));
        }
        float e[DPL], den = 0.0f;
#pragma unroll
        for (int d = 0; d < DPL; ++d) e[d] = expf(mv::sub_rn(score[d], mx));
#pragma unroll
        for (int j = 0; j < QL; ++j)
#pragma unroll
            for (int d = 0; d < DPL; ++d)
                den = mv::add_rn(den, QL == 1 ? e[d] : __shfl(e[d], (j * PPW + pl) * LPP + sub));
        const mv::Recip rden = mv::make_recip(den);
        const float wmax = mv::div_rn(1.0f, rden);   // attn_fuse_d = False: max_d softmax = exp(0) / sum
#pragma unroll
        for (int d = 0; d < DPL; ++d) {
            const float wgt = a.fuse_d ? mv::div_rn(mv::div_rn(e[d], rden), sqrt_c) : wmax;
            wsum[d] = mv::add_rn(wsum[d], wgt);
#pragma unroll
            for (int k = 0; k < GPL; ++k) acc[d][k] = mv::add_rn(acc[d][k], mv::mul_rn(wgt, cg[d][k]));
        }
    }

    if (valid) {
#pragma unroll
        for (int d = 0; d < DPL; ++d) {
            const long o = (((long)b * D + qg * DPL + d) * hw + p);
            float* op = a.out + o * G + sub * GPL;
            const mv::Recip rw = mv::make_recip(wsum[d]);
            if (GPL == 4) {
                st4(op, (f32x4){mv::div_rn(acc[d][0], rw), mv::div_rn(acc[d][GPL > 1 ? 1 : 0], rw),
                                mv::div_rn(acc[d][GPL > 2 ? 2 : 0], rw), mv::div_rn(acc[d][GPL > 3 ? 3 : 0], rw)});
            } else {
#pragma unroll
                for (int k = 0; k < GPL; ++k) op[k] = mv::div_rn(acc[d][k], rw);
            }
            if (a.wsum_out && sub == 0) a.wsum_out[o] = wsum[d];
        }
    }
}

// launch shape of the pixel-major kernel: hypotheses per lane (0 = all D) and the occupancy target handed to the
// register allocator; MVSTER_PIX_DPL / MVSTER_PIX_WPE override the defaults for experiments
[[maybe_unused]] static const int g_pix_dpl = MV_PROBE_ENV("MVSTER_PIX_DPL") ? atoi(MV_PROBE_ENV("MVSTER_PIX_DPL")) : 2;
[[maybe_unused]] static const int g_pix_wpe = MV_PROBE_ENV("MVSTER_PIX_WPE") ? atoi(MV_PROBE_ENV("MVSTER_PIX_WPE")) : 4;

template <int C, int G, int D, int DPL, int WPE>
int launch_fwd_pix_cfg(const WarpAggArgs& a, hipStream_t stream) {
    constexpr int PPW = 64 / ((C / 8) * (D / DPL));
    dim3 grid((a.h * a.w + PPW - 1) / PPW, a.B);
    MV_NOTE_KERNEL("warp_agg_fwd_pix_kernel<%d, %d, %d, %d, %d>", C, G, D, DPL, WPE);
    hipLaunchKernelGGL((warp_agg_fwd_pix_kernel<C, G, D, DPL, WPE>), grid, dim3(64), 0, stream, a);
    return mv_check_launch();
}

template <int C, int G, int D>
int launch_fwd_pix(const WarpAggArgs& a, hipStream_t stream) {
    // 24-bit multiply-add on texel indices, 32-bit byte offsets inside one (view, batch) map
    if ((long)a.Hs * a.Ws >= (1L << 23) || (long)a.Hs * a.Ws * C * 4 >= (1L << 31)) return MVSTER_ERR_SHAPE;
    if (g_pix_dpl == 2 || (C / 8) * (D / 2) > 64) {
        if constexpr ((C / 8) * (D / 2) <= 64) {
            if (g_pix_wpe >= 6) return launch_fwd_pix_cfg<C, G, D, 2, 6>(a, stream);
            if (g_pix_wpe == 5) return launch_fwd_pix_cfg<C, G, D, 2, 5>(a, stream);
            return launch_fwd_pix_cfg<C, G, D, 2, 4>(a, stream);
        }
    }
    if (g_pix_wpe >= 5) return launch_fwd_pix_cfg<C, G, D, D, 5>(a, stream);
    return launch_fwd_pix_cfg<C, G, D, D, 4>(a, stream);
}

template <int C, int G>
int dispatch_fwd_pix(const WarpAggArgs& a, hipStream_t stream) {
    if (a.D == 4) return launch_fwd_pix<C, G, 4>(a, stream);
    if (a.D == 8) return launch_fwd_pix<C, G, 8>(a, stream);
    return MVSTER_ERR_UNSUPPORTED;
}

#endif  // MVSTER_PROBES

template <int C, int G, int D>
int launch_fwd_wave(const WarpAggArgs& a, hipStream_t stream, int sched = 0) {
    constexpr int PPB = 4 * (64 / ((C / 8) * D));
    // 24-bit multiply-add on texel indices, 32-bit byte offsets inside one (view, batch) map
    if ((long)a.Hs * a.Ws >= (1L << 23) || (long)a.Hs * a.Ws * C * 4 >= (1L << 31)) return MVSTER_ERR_SHAPE;
    dim3 grid((a.h * a.w + PPB - 1) / PPB, a.B);
    if (sched == 1) {
        MV_NOTE_KERNEL("warp_agg_fwd_wave_kernel<%d, %d, %d, 1>", C, G, D);
        hipLaunchKernelGGL((warp_agg_fwd_wave_kernel<C, G, D, 1>), grid, dim3(256), 0, stream, a);
    } else if (sched == 2) {
        MV_NOTE_KERNEL("warp_agg_fwd_wave_kernel<%d, %d, %d, 2>", C, G, D);
        hipLaunchKernelGGL((warp_agg_fwd_wave_kernel<C, G, D, 2>), grid, dim3(256), 0, stream, a);
    } else {
        MV_NOTE_KERNEL("warp_agg_fwd_wave_kernel<%d, %d, %d>", C, G, D);
        hipLaunchKernelGGL((warp_agg_fwd_wave_kernel<C, G, D>), grid, dim3(256), 0, stream, a);
    }
    return mv_check_launch();
}

#ifdef MVSTER_PROBES   // LDS-window form (variant 5): probe build only
// ------------------------------------------------------------------------------------------
// LDS-staged source windows (variant 5; C in {8, 16}: the two fine stages, where the time is).
// The wave-local kernel gathers every tap through the texture path: 4 taps x 32 bytes per (pixel, d, view), 671 MB
// through L1 per stage-4 launch against 80 MB of HBM traffic, texture-address unit 55 % busy, and inside the forward
// (features not cache-resident) it waits on many small dependent misses: 48 us against 30 us warm.  Here a workgroup
// owns a TW x TH tile of reference pixels and walks it in NP passes with the wave kernel's lane mapping; per view it
//   A. projects all its (pixel, d) pairs and reduces the bounding box of the taps that carry weight (wave butterfly +
//      four LDS atomics),
//   B. stages that source window once -- whole texel rows are contiguous in channels-last memory, so this is a handful
//      of fully coalesced 16-byte loads per thread -- into LDS, one plane per 16-byte channel quad (conflict-free reads),
//   C. serves the taps with ds_read_b128.  A lane whose weighted taps do not fit the window (capacity WCAP texels:
//      geometrically incoherent hypotheses) takes the buffer-load path of the wave kernel instead, lane by lane.
// Arithmetic, lane mapping and summation order are the wave kernel's: bit-identical output (a zero-weight tap may read
// another in-window texel: 0 * finite = 0 either way).  Pays off when neighbouring pixels carry similar depths (trained
// networks, stage 1 by construction); on unrelated winner-take-all depths most lanes fall back and the bounding-box
// pass is pure overhead -- the plan keeps the wave kernel there.
// ------------------------------------------------------------------------------------------
#ifndef MV_TILE_WAVES
#define MV_TILE_WAVES 3      // (measured: 2 -> 66.6 us, 3 -> 57.6 us, 4 -> 77.5 us at stage 4, smooth depths; the wave kernel: 32.3 us)
#endif
template <int C, int G, int D, int TW, int TH>
__global__ void __launch_bounds__(256, MV_TILE_WAVES) warp_agg_fwd_tile_kernel(WarpAggArgs a, int tiles_x, int tiles_y) {
    constexpr int LPP = C / 8, CG = C / G, GPL = 8 / CG, PPW = 64 / (LPP * D);
    constexpr int NP = (TW * TH) / (4 * PPW);                     // passes over the tile
    constexpr int NQ = C / 4;                                     // 16-byte quads per texel = LDS planes
    constexpr int WCAP = 24 * 1024 / (C * 4);                     // window capacity in texels (24 KB)
    static_assert(C % 8 == 0 && CG <= 8 && 8 % CG == 0 && PPW * LPP * D == 64 && NP * 4 * PPW == TW * TH, "tile split");
    __shared__ f32x4 win[NQ * WCAP];
    __shared__ int box[4];                                        // min x, min y, max x, max y of the weighted taps

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPP, pl = (lane / LPP) % PPW, d = lane / (LPP * PPW);
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ty0 = (int)(tile / (unsigned)tiles_x) * TH, tx0 = (int)(tile % (unsigned)tiles_x) * TW;
    const mv::GridNorm gn = mv::make_grid_norm(a.Hs, a.Ws);
    const mv::Recip temp = mv::make_recip(a.attn_temp), sqrt_c = mv::make_recip(a.sqrt_c);
    constexpr int SH = C == 8 ? 5 : 6;                            // log2(bytes per texel)
    const float xhi = (float)(a.Ws + 4), yhi = (float)(a.Hs + 4);
    const unsigned src_bytes = (unsigned)a.Hs * (unsigned)a.Ws * (unsigned)(C * 4);
    const int row_bytes = a.Ws << SH;

    // per pass: pixel, reference features, hypothesis
    int pix[NP];
    bool valid[NP];
    float depth[NP];
    f32x4 R0[NP], R1[NP];
    float acc[NP][GPL], wsum[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = (k * 4 + wave) * PPW + pl;                  // tile-local pixel index, row-major
        const int y = ty0 + q / TW, x = tx0 + q % TW;
        valid[k] = y < a.h && x < a.w;
        pix[k] = valid[k] ? y * a.w + x : hw - 1;                 // clamped: every lane takes part in the shuffles
        depth[k] = a.hypo[((long)b * D + d) * hw + pix[k]];
        const float* rp = a.ref + (long)b * a.ref_bs + (long)pix[k] * C + sub * 8;
        R0[k] = ld4(rp);
        R1[k] = ld4(rp + 4);
#pragma unroll
        for (int g = 0; g < GPL; ++g) acc[k][g] = 0.0f;
        wsum[k] = 1e-8f;
    }

    for (int v = 0; v < a.NV; ++v) {
        mv::RT m;
        const float* r = a.rt + ((long)b * a.NV + v) * 12;        // wave-uniform
#pragma unroll
        for (int i = 0; i < 9; ++i) m.r[i] = r[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) m.t[i] = r[9 + i];
        if (threadIdx.x == 0) { box[0] = 1 << 30; box[1] = 1 << 30; box[2] = -1; box[3] = -1; }
        // ---- A: taps of every pass, bounding box of the ones that carry weight
        mv::Taps t[NP];
        int bx0 = 1 << 30, by0 = 1 << 30, bx1 = -1, by1 = -1;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int y = pix[k] / a.w, x = pix[k] - y * a.w;
            float sx, sy;
            mv::project(m, (float)x, (float)y, depth[k], gn, sx, sy);
            const float cx = __builtin_amdgcn_fmed3f(sx, -4.0f, xhi), cy = __builtin_amdgcn_fmed3f(sy, -4.0f, yhi);
            const float fx = floorf(cx), fy = floorf(cy);
            t[k].x0 = (int)fx;
            t[k].y0 = (int)fy;
            const float wx1 = mv::sub_rn(cx, fx), wy1 = mv::sub_rn(cy, fy);
            const float wx0 = mv::sub_rn(1.0f, wx1), wy0 = mv::sub_rn(1.0f, wy1);
            const bool vx0 = (unsigned)t[k].x0 < (unsigned)a.Ws, vx1 = (unsigned)(t[k].x0 + 1) < (unsigned)a.Ws;
            const bool vy0 = (unsigned)t[k].y0 < (unsigned)a.Hs, vy1 = (unsigned)(t[k].y0 + 1) < (unsigned)a.Hs;
            t[k].nw = (vy0 && vx0) ? mv::mul_rn(wy0, wx0) : 0.0f;
            t[k].ne = (vy0 && vx1) ? mv::mul_rn(wy0, wx1) : 0.0f;
            t[k].sw = (vy1 && vx0) ? mv::mul_rn(wy1, wx0) : 0.0f;
            t[k].se = (vy1 && vx1) ? mv::mul_rn(wy1, wx1) : 0.0f;
            if (valid[k] && (vx0 || vx1) && (vy0 || vy1)) {       // some tap lies inside the map
                bx0 = min(bx0, max(t[k].x0, 0));
                by0 = min(by0, max(t[k].y0, 0));
                bx1 = max(bx1, min(t[k].x0 + 1, a.Ws - 1));
                by1 = max(by1, min(t[k].y0 + 1, a.Hs - 1));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            bx0 = min(bx0, __shfl_xor(bx0, off));
            by0 = min(by0, __shfl_xor(by0, off));
            bx1 = max(bx1, __shfl_xor(bx1, off));
            by1 = max(by1, __shfl_xor(by1, off));
        }
        __syncthreads();                                          // box initialised; previous view's window no longer read
        if (lane == 0 && bx1 >= 0) {
            atomicMin(&box[0], bx0);
            atomicMin(&box[1], by0);
            atomicMax(&box[2], bx1);
            atomicMax(&box[3], by1);
        }
        __syncthreads();
        // window = the box, cut to the capacity (anchored at its top-left corner); an empty box -> texel (0, 0)
        int wx0 = box[0], wy0 = box[1], wx = box[2] - wx0 + 1, wy = box[3] - wy0 + 1;
        if (box[2] < 0) { wx0 = 0; wy0 = 0; wx = 1; wy = 1; }
        wx = min(wx, WCAP);
        wy = min(wy, WCAP / wx);
        // ---- B: stage the window: thread -> (texel, quad), a window row is one contiguous run of wx * C * 4 bytes
        const float* sp = a.src + (long)v * a.src_vs + (long)b * a.src_bs;
        const int nst = wy * wx * NQ;
        for (int i = threadIdx.x; i < nst; i += 256) {
            const int qd = i % NQ, tx = (i / NQ) % wx, tyy = i / (NQ * wx);
            win[qd * WCAP + tyy * wx + tx] = ld4(sp + ((long)(wy0 + tyy) * a.Ws + (wx0 + tx)) * C + qd * 4);
        }
        __syncthreads();
        // ---- C: the wave kernel's arithmetic, taps from LDS where the lane's weighted taps lie inside the window
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sp), (short)0, (int)src_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const mv::Taps& tk = t[k];
            // in-window test on the clamped extent of the taps (taps outside the map carry no weight)
            const int ex0 = max(tk.x0, 0), ey0 = max(tk.y0, 0), ex1 = min(tk.x0 + 1, a.Ws - 1), ey1 = min(tk.y0 + 1, a.Hs - 1);
            const bool weighted = (ex0 <= ex1) && (ey0 <= ey1);
            const bool inwin = !weighted || (ex0 >= wx0 && ey0 >= wy0 && ex1 < wx0 + wx && ey1 < wy0 + wy);
            f32x4 q0, q1, q2, q3, q4, q5, q6, q7;
            if (inwin) {
                // (a tap outside the window has zero weight: it may read any staged texel)
                const int rx0 = min(max(tk.x0 - wx0, 0), wx - 1), rx1 = min(max(tk.x0 + 1 - wx0, 0), wx - 1);
                const int ry0 = min(max(tk.y0 - wy0, 0), wy - 1), ry1 = min(max(tk.y0 + 1 - wy0, 0), wy - 1);
                const f32x4* w0 = win + (sub * 2) * WCAP;
                const f32x4* w1 = w0 + WCAP;
                q0 = w0[ry0 * wx + rx0]; q1 = w1[ry0 * wx + rx0];
                q2 = w0[ry0 * wx + rx1]; q3 = w1[ry0 * wx + rx1];
                q4 = w0[ry1 * wx + rx0]; q5 = w1[ry1 * wx + rx0];
                q6 = w0[ry1 * wx + rx1]; q7 = w1[ry1 * wx + rx1];
            } else {
                const unsigned oa = (((unsigned)__mul24(tk.y0, a.Ws) + (unsigned)tk.x0) << SH) + (unsigned)(sub * 32);
                const unsigned ob = oa + (unsigned)row_bytes;
                q0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa, 0, 0));
                q1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + 16, 0, 0));
                q2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + (4 * C), 0, 0));
                q3 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, oa + (4 * C + 16), 0, 0));
                q4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob, 0, 0));
                q5 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + 16, 0, 0));
                q6 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + (4 * C), 0, 0));
                q7 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ob + (4 * C + 16), 0, 0));
            }
            float part[GPL];
#define MV_PAIR(c, A, B, Cq, Dq, R, j)                                                                         \
            {                                                                                                  \
                const mv::f32x2 wv = mv::blend2(tk.nw, tk.ne, tk.sw, tk.se, (mv::f32x2){A[j], A[j + 1]},       \
                                                (mv::f32x2){B[j], B[j + 1]}, (mv::f32x2){Cq[j], Cq[j + 1]},    \
                                                (mv::f32x2){Dq[j], Dq[j + 1]});                                \
                const mv::f32x2 p2 = mv::mul_rn2(wv, (mv::f32x2){R[j], R[j + 1]});                             \
                part[(c) / CG] = ((c) % CG == 0) ? p2[0] : mv::add_rn(part[(c) / CG], p2[0]);                  \
                part[((c) + 1) / CG] = (((c) + 1) % CG == 0) ? p2[1] : mv::add_rn(part[((c) + 1) / CG], p2[1]); \
            }
            MV_PAIR(0, q0, q2, q4, q6, R0[k], 0)
            MV_PAIR(2, q0, q2, q4, q6, R0[k], 2)
            MV_PAIR(4, q1, q3, q5, q7, R1[k], 0)
            MV_PAIR(6, q1, q3, q5, q7, R1[k], 2)
#undef MV_PAIR
            float cg[GPL];
#pragma unroll
            for (int g = 0; g < GPL; ++g) cg[g] = mv::div_rn(part[g], (float)CG);   // .mean(2)
            float score = 0.0f;
            const int lane0 = lane - sub;
#pragma unroll
            for (int j = 0; j < LPP; ++j)
#pragma unroll
                for (int g = 0; g < GPL; ++g) {
                    const float val = LPP == 1 ? cg[g] : __shfl(cg[g], lane0 + j);
                    score = (j == 0 && g == 0) ? val : mv::add_rn(score, val);
                }
            if (a.fuse_d) score = mv::div_rn(score, temp);
            float mx = score;
#pragma unroll
            for (int j = 0; j < D; ++j) mx = fmaxf(mx, __shfl(score, (j * PPW + pl) * LPP + sub));
            const float e = expf(mv::sub_rn(score, mx));
            float den = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) den = mv::add_rn(den, __shfl(e, (j * PPW + pl) * LPP + sub));
            const mv::Recip rden = mv::make_recip(den);
            float wgt;
            if (a.fuse_d)
                wgt = mv::div_rn(mv::div_rn(e, rden), sqrt_c);
            else
                wgt = mv::div_rn(1.0f, rden);
            wsum[k] = mv::add_rn(wsum[k], wgt);
#pragma unroll
            for (int g = 0; g < GPL; ++g) acc[k][g] = mv::add_rn(acc[k][g], mv::mul_rn(wgt, cg[g]));
        }
    }

#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (!valid[k]) continue;
        const long o = (((long)b * D + d) * hw + pix[k]);
        float* op = a.out + o * G + sub * GPL;
        const mv::Recip rw = mv::make_recip(wsum[k]);
        if (GPL == 4) {
            st4(op, (f32x4){mv::div_rn(acc[k][0], rw), mv::div_rn(acc[k][GPL > 1 ? 1 : 0], rw),
                            mv::div_rn(acc[k][GPL > 2 ? 2 : 0], rw), mv::div_rn(acc[k][GPL > 3 ? 3 : 0], rw)});
        } else {
#pragma unroll
            for (int g = 0; g < GPL; ++g) op[g] = mv::div_rn(acc[k][g], rw);
        }
        if (a.wsum_out && sub == 0) a.wsum_out[o] = wsum[k];
    }
}

template <int C, int G, int D>
int launch_fwd_tile(const WarpAggArgs& a, hipStream_t stream) {
    constexpr int TW = 32, TH = C == 8 ? 8 : 4;
    if ((long)a.Hs * a.Ws >= (1L << 23) || (long)a.Hs * a.Ws * C * 4 >= (1L << 31)) return MVSTER_ERR_SHAPE;
    const int tiles_x = (a.w + TW - 1) / TW, tiles_y = (a.h + TH - 1) / TH;
    MV_NOTE_KERNEL("warp_agg_fwd_tile_kernel<%d, %d, %d, %d, %d>", C, G, D, TW, TH);
    hipLaunchKernelGGL((warp_agg_fwd_tile_kernel<C, G, D, TW, TH>), dim3(tiles_x * tiles_y, a.B), dim3(256), 0, stream, a, tiles_x,
                       tiles_y);
    return mv_check_launch();
}

template <int C, int G>
int dispatch_fwd_tile(const WarpAggArgs& a, hipStream_t stream) {
    if (a.D == 4) return launch_fwd_tile<C, G, 4>(a, stream);
    if (a.D == 8) return launch_fwd_tile<C, G, 8>(a, stream);
    return MVSTER_ERR_UNSUPPORTED;
}

#endif  // MVSTER_PROBES

template <int C, int G>
int dispatch_fwd_wave(const WarpAggArgs& a, hipStream_t stream, int sched = 0) {
    if (a.D == 4) return launch_fwd_wave<C, G, 4>(a, stream, sched);
    if (a.D == 8) return launch_fwd_wave<C, G, 8>(a, stream, sched);
    return MVSTER_ERR_UNSUPPORTED;
}

template <int C, int G>
int launch_fwd_lanes(const WarpAggArgs& a, hipStream_t stream) {
    constexpr int PPB = 64 / (C / 8);
    if (a.D > 8) return MVSTER_ERR_UNSUPPORTED;
    dim3 block(64, a.D);
    dim3 grid((a.h * a.w + PPB - 1) / PPB, a.B);
    MV_NOTE_KERNEL("warp_agg_fwd_lanes_kernel<%d, %d, 8>", C, G);
    hipLaunchKernelGGL((warp_agg_fwd_lanes_kernel<C, G, 8>), grid, block, 0, stream, a);
    return mv_check_launch();
}

template <int C, int G, bool GROUP>
int launch_fwd(const WarpAggArgs& a, hipStream_t stream) {
    dim3 block(64, a.D);
    dim3 grid((a.h * a.w + 63) / 64, a.B);
    if (a.D <= 8) {
        MV_NOTE_KERNEL("warp_agg_fwd_kernel<%d, %d, %s, 8>", C, G, GROUP ? "true" : "false");
        hipLaunchKernelGGL((warp_agg_fwd_kernel<C, G, GROUP, 8>), grid, block, 0, stream, a);
    } else {
        // 1024-thread blocks; the per-thread correlations of the widest ungrouped case do not fit LDS
        if constexpr (G * kMaxD * 64 * 4 > 120 * 1024) return MVSTER_ERR_UNSUPPORTED;
        else if (a.D <= kMaxD) {
            MV_NOTE_KERNEL("warp_agg_fwd_kernel<%d, %d, %s, %d>", C, G, GROUP ? "true" : "false", kMaxD);
            hipLaunchKernelGGL((warp_agg_fwd_kernel<C, G, GROUP, kMaxD>), grid, block, 0, stream, a);
        } else if (a.D <= 32) {          // the same 1024 threads as 32 pixels x 32 hypotheses
            MV_NOTE_KERNEL("warp_agg_fwd_kernel<%d, %d, %s, 32, 32>", C, G, GROUP ? "true" : "false");
            hipLaunchKernelGGL((warp_agg_fwd_kernel<C, G, GROUP, 32, 32>), dim3((a.h * a.w + 31) / 32, a.B), dim3(32, a.D), 0,
                               stream, a);
        } else {                         // ... 16 pixels x 64 hypotheses
            MV_NOTE_KERNEL("warp_agg_fwd_kernel<%d, %d, %s, 64, 16>", C, G, GROUP ? "true" : "false");
            hipLaunchKernelGGL((warp_agg_fwd_kernel<C, G, GROUP, 64, 16>), dim3((a.h * a.w + 15) / 16, a.B), dim3(16, a.D), 0,
                               stream, a);
        }
    }
    return mv_check_launch();
}

// ------------------------------------------------------------------------------------------
// Backward w.r.t. the features (training; autograd of mvs4net_utils.py:1036-1060).  The grid is
// not differentiated (torch.no_grad at :23).  Per (pixel, d) thread and per view: re-gather the
// taps, redo the depth softmax, form dL/dcor, then re-gather once more to scatter
//   d src[tap][c] += w_tap * dwarp[c]      (atomic, 4 taps x C)
//   d ref[c]      += ...                   (atomic, summed over d and views)
// Uses `out` and `wsum` saved by the forward.  Both attention forms (attn_fuse_d on: a weight per (pixel, d); off: one
// per pixel, the largest softmax value along depth, mvs4net_utils.py:1048-1051), up to 16 hypotheses.
// ------------------------------------------------------------------------------------------
// LDS accumulators of the backward are 64-bit FIXED POINT: on gfx950 a wave64 ds_add_f32 costs ~190 LDS cycles
// (scripts/probes/lds_atomic_probe.hip: it is executed lane by lane), ds_add_u64 ~7.  The scatter window and the
// reference-gradient tile therefore accumulate round(v * 2^k) with integer atomics -- which are also associative, so the
// sums no longer depend on the order in which the waves arrive -- and are converted back once.  2^k is chosen per
// launch from upper bounds on the operands (the largest |grad_out|, |ref|, |src|, reduced on the device just before):
// the largest single contribution lands near 2^36, leaving 27 bits of head room for the number of contributions per
// texel and >= 29 bits below the largest value actually seen (fp32 atomics keep 24).
typedef unsigned long long u64;

struct FixScale {
    float s, inv;
};

// maxima = {max |grad_out|, max |ref|, max |src|}
__device__ __forceinline__ FixScale make_fix_scale(const float* maxima, int G, int D, int CG, bool group, bool fuse, float temp) {
    const float gomax = maxima[0], fmax = fmaxf(maxima[1], maxima[2]);
    // |cor| <= cormax; |d L / d cor| <= gomax * (1 + 2 G (1 + D) cormax / temp)  (softmax Jacobian: |sig * dsig| <= 2 G gomax
    // cormax because W >= the view's own weight); a tap contribution is that times a feature value (/ CG), weights <= 1
    const float cormax = group ? fmax * fmax : 4.0f * fmax * fmax;
    const float dcor = gomax * (1.0f + 2.0f * (float)G * (float)(1 + D) * cormax / (fuse ? temp : 1.0f));
    const float bound = group ? dcor * fmax / (float)CG : 4.0f * fmax * dcor;
    FixScale f;
    int e = 0;
    if (bound > 0.0f && bound < INFINITY) frexpf(bound, &e);      // bound < 2^e
    e = min(max(36 - e, -60), 100);
    f.s = ldexpf(1.0f, e);
    f.inv = ldexpf(1.0f, -e);
    // A non-finite operand (absmax_kernel reports NaN for it) has no fixed-point image: integer conversion would turn it
    // into finite garbage.  Every value converted back then reads NaN instead -- the gradients of such a step are NaN, like
    // the reference's, never silently finite.  (Resolution otherwise: absolute, bound * 2^-36 per contribution, i.e. the
    // smallest gradients of a launch keep fewer bits than its largest; fp32 atomics would keep 24 relative to the running sum.)
    if (!(bound < INFINITY)) { f.s = 0.0f; f.inv = __builtin_nanf(""); }
    return f;
}
// round-to-nearest-even float -> 64-bit integer for |t| < 2^51 in four instructions (the library conversion takes ~12): adding
// 1.5 * 2^52 in double leaves round(t) in the low mantissa bits, two's complement included
__device__ __forceinline__ u64 fix_cvt(float t) {
    const double d = (double)t + 6755399441055744.0;
    return (u64)(__double_as_longlong(d) - 0x4338000000000000LL);
}
__device__ __forceinline__ void fix_add(u64* p, float v, float s) { atomicAdd(p, fix_cvt(v * s)); }
__device__ __forceinline__ float fix_get(u64 v, float inv) { return (float)(long long)v * inv; }

// max |x| of up to three arrays in one launch: workgroups [first[k], first[k+1]) reduce x[k] (n[k] floats, 16-byte aligned)
// into out[k] (a non-negative float orders like its bit pattern); out[] must start at 0.  ONE atomic per workgroup and at
// most ~1.5k workgroups per launch: atomics on one address retire ~13 ns apart on gfx950, and with one per wave and a launch
// per array (47 k atomics per training step) the twelve launches took 590 us for 262 MB.
struct AbsMaxArgs {
    const float* x[3];
    long n[3];
    int first[4];
};
constexpr int kAbsMaxBlocks = 1536;

__device__ __forceinline__ void absmax_block(const AbsMaxArgs& a, float* __restrict__ out, int b) {
    __shared__ float wave_max[4];
    __shared__ int wave_bad[4];
    const int k = (b >= a.first[1] ? 1 : 0) + (b >= a.first[2] ? 1 : 0);
    const float* __restrict__ x = a.x[k];
    const long n = a.n[k];
    float m = 0.0f;
    bool bad = false;           // fmaxf drops NaN: non-finite elements are tracked separately and reported as NaN
    const long n4 = n >> 2;
    const long stride = (long)(a.first[k + 1] - a.first[k]) * 256;
    auto take = [&](const f32x4& v) {
        const float a0 = fabsf(v[0]), a1 = fabsf(v[1]), a2 = fabsf(v[2]), a3 = fabsf(v[3]);
        const float mv = fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
        m = fmaxf(m, mv);
        // (fmaxf drops a NaN operand: the sum below is NaN / Inf exactly if one of the four is)
        bad = bad || !((a0 + a1) + (a2 + a3) < INFINITY);
    };
    const long t0 = (long)(b - a.first[k]) * 256 + threadIdx.x;
    long i = t0;
    // four independent 16-byte loads in flight per thread
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const f32x4 v0 = ld4(x + i * 4), v1 = ld4(x + (i + stride) * 4), v2 = ld4(x + (i + 2 * stride) * 4),
                    v3 = ld4(x + (i + 3 * stride) * 4);
        take(v0);
        take(v1);
        take(v2);
        take(v3);
    }
    for (; i < n4; i += stride) take(ld4(x + i * 4));
    for (long j = (n4 << 2) + t0; j < n; j += stride) {
        m = fmaxf(m, fabsf(x[j]));
        bad = bad || !(fabsf(x[j]) < INFINITY);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    const bool any_bad = __any(bad);
    if ((threadIdx.x & 63) == 0) { wave_max[threadIdx.x >> 6] = m; wave_bad[threadIdx.x >> 6] = any_bad ? 1 : 0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
        const bool wg_bad = (wave_bad[0] | wave_bad[1] | wave_bad[2] | wave_bad[3]) != 0;
        // (as integers, positive floats order like their values and the quiet-NaN pattern lies above +Inf)
        if (m > 0.0f || wg_bad) atomicMax(reinterpret_cast<int*>(out + k), wg_bad ? 0x7fc00000 : __float_as_int(m));
    }
}

__global__ void __launch_bounds__(256) absmax_kernel(AbsMaxArgs a, float* __restrict__ out) { absmax_block(a, out, blockIdx.x); }

// the workgroup layout of launch_absmax, for kernels that carry the reduction beside other work
static AbsMaxArgs absmax_args(const float* const* x, const long* n, int count) {
    AbsMaxArgs a;
    long total = 0;
    for (int k = 0; k < count; ++k) total += n[k];
    a.first[0] = 0;
    for (int k = 0; k < 3; ++k) {
        a.x[k] = k < count ? x[k] : nullptr;
        a.n[k] = k < count ? n[k] : 0;
        long nb = 0;
        if (k < count) {
            nb = total > 0 ? (long)kAbsMaxBlocks * n[k] / total : 1;
            nb = std::min(nb, n[k] / 4096 + 1);
            nb = std::max(nb, 1L);
        }
        a.first[k + 1] = a.first[k] + (int)nb;
    }
    return a;
}

// workgroups in proportion to the arrays' sizes (they finish together), each array at least one
void launch_absmax(const float* const* x, const long* n, int count, float* out, hipStream_t s) {
    const AbsMaxArgs a = absmax_args(x, n, count);
    // (unused entries own no workgroups: first[k] == first[k+1] == gridDim.x, never selected)
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)a.first[3]), dim3(256), 0, s, a, out);
}

struct WarpAggBwdArgs {
    WarpAggArgs f;
    const float* fwd_out;   // [B, D, h, w, G]
    const float* wsum;      // [B, D, h, w]
    const float* grad_out;  // [B, D, h, w, G]
    float* grad_ref;        // [B, h, w, C]
    float* grad_src;        // [NV][B, Hs, Ws, C]  (same strides as src)
    // Deterministic, atomic-free accumulation of grad_src (both null: the window is flushed with global atomics):
    // every workgroup stores its scatter windows densely, windows [B][nblk][NV][C/8][WY][kWinX][8] and their origins
    // win_org [B][nblk][NV][2]; scatter_gather_kernel then sums, per source texel, the windows that cover it.
    float* windows;
    int* win_org;
    const float* maxima;    // {max |grad_out|, max |ref|, max |src|} (device; written just before this launch)
    // sorted scatter (mvster_warp_agg_bwd_sorted; REC instances of warp_agg_bwd_kernel): see the section below
    const int* rec_offset;  // [B*NV*ntiles + 1] exclusive prefix sums of the tiles' record counts
    int* rec_cursor;        // [B*NV*ntiles] running cursors (zero before the launch)
    float* rec;             // the records, kRecWords floats each
    int tiles_x, tiles_y;   // source tiles of kRecTileX x kRecTileY texels
};

// ------------------------------------------------------------------------------------------
// Sorted scatter of the source-view gradient (the adjoint of grid_sample, mvs4net_utils.py:13-59) WITHOUT global atomics.
// A (pixel, hypothesis, view) sample adds w_tap * dw[c] to four source texels.  With unrelated hypotheses on neighbouring
// pixels (random-winner depth maps) those texels are spread over hundreds of pixels of an epipolar line: the scatter
// windows above catch little and the rest are random-address global atomics, which the chip retires at ~370 G adds/s
// whatever the kernel does (0.92 ms for the 335 M adds of the full-resolution stage).  Here the samples are counting-sorted
// by SOURCE tile instead:
//   K0 warp_bwd_count_kernel   projects every sample and counts, per (batch, view, 32x32 source tile), the samples with a
//                              weighted tap in the tile (a sample on a tile border counts in up to four tiles)
//   K-scan                     exclusive prefix sums of the counts
//   K1 warp_agg_bwd_kernel<.., REC = true>   the backward's arithmetic as before, but instead of scattering it appends a
//                              48-byte record {corner, tap mask, east / south fractions, dw[8]} per 8-channel block to the
//                              list of every tile the sample touches (slots reserved per workgroup: one returning global
//                              atomic per touched tile and view, not per sample)
//   K2 warp_bwd_accum_kernel   one workgroup per (batch, view, channel block, tile): accumulates the tile's records in a
//                              64-bit fixed-point LDS window (integer adds: the order of the records does not matter) and
//                              writes the tile with plain coalesced stores -- every texel of grad_src exactly once, so
//                              the buffer needs no zero fill either.
// Deterministic by construction and independent of how smooth the depth maps are.
// ------------------------------------------------------------------------------------------
constexpr int kRecTileX = 32, kRecShiftX = 5, kRecTileY = 16, kRecShiftY = 4;     // source tile: 32 x 16 texels
constexpr int kRecWords = 12;            // {packed corner + mask, wx1, wy1, 0}, dw[0..3], dw[4..7]
constexpr int kRecMaxTiles = 4096;       // source tiles per map the workgroups' LDS histograms hold (4096 x 512 texels)

// bit k set: tap k carries weight (k = 0 nw (x0, y0), 1 ne (x0+1, y0), 2 sw (x0, y0+1), 3 se); t after clamp_taps()
__device__ __forceinline__ unsigned tap_mask(const mv::Taps& t) {
    return (t.nw != 0.0f ? 1u : 0u) | (t.ne != 0.0f ? 2u : 0u) | (t.sw != 0.0f ? 4u : 0u) | (t.se != 0.0f ? 8u : 0u);
}

// The tiles that hold a weighted tap of the sample: tile[k] = tile of tap k, or -1 where the tap carries no weight or an
// earlier tap already named the tile (statically indexed throughout: no compaction, nothing for the compiler to spill).
struct SampleTiles {
    int tile[4];
};

__device__ __forceinline__ SampleTiles sample_tiles(const mv::Taps& t, unsigned mask, int tiles_x) {
    SampleTiles s;
    const int tx0 = t.x0 >> kRecShiftX, tx1 = (t.x0 + 1) >> kRecShiftX;
    const int ty0 = t.y0 >> kRecShiftY, ty1 = (t.y0 + 1) >> kRecShiftY;
    const int id0 = ty0 * tiles_x + tx0, id1 = ty0 * tiles_x + tx1, id2 = ty1 * tiles_x + tx0, id3 = ty1 * tiles_x + tx1;
    s.tile[0] = (mask & 1u) ? id0 : -1;
    s.tile[1] = ((mask & 2u) && id1 != s.tile[0]) ? id1 : -1;
    s.tile[2] = ((mask & 4u) && id2 != s.tile[0] && id2 != s.tile[1]) ? id2 : -1;
    s.tile[3] = ((mask & 8u) && id3 != s.tile[0] && id3 != s.tile[1] && id3 != s.tile[2]) ? id3 : -1;
    return s;
}

// Scatter window: the source-view gradient of one workgroup (64 reference pixels of a row x all depths) and
// one view lands on a compact patch of the source map (a few rows around an epipolar segment), so it is
// accumulated in LDS (ds_add_f32) over a kWinX x kWinY texel window anchored at the workgroup's smallest tap
// coordinates, 8 channels at a time, and flushed with ONE global atomic per touched (texel, channel) instead
// of one per (pixel, depth, tap, channel): 5-8x fewer global atomics, which is what bounds this kernel.  Taps
// that fall outside the window (strongly rotated views) go to global memory directly.  grad_ref needs no
// atomics at all: each reference pixel belongs to exactly one workgroup, which sums over depths and views in
// LDS and stores once.
// Taps that fall outside the scatter window go to global memory with atomics.  Issued lane = pixel they would be 8
// successive instructions per tap, each lane walking its own texel's channels: the L2 atomic units then see one 4-byte
// request per lane and instruction, ~20 G adds/s whatever the address pattern (scripts/probes/global_atomic_probe.hip).
// Eight neighbouring lanes covering one texel's 32 contiguous bytes are merged into one request: 169 G adds/s.  So a
// wavefront parks the (offset, 8 values) records of its out-of-window lanes in LDS and drains them with
// lane = (record, channel).  Wave-local: no workgroup barrier, 64 records of 36 bytes per wavefront.
struct TapQueue {
    int off[64];
    float val[64][8];
};
constexpr int kQueueDirect = 4;      // up to this many out-of-window lanes of a wavefront issue their atomics themselves

__device__ __forceinline__ void queue_tap(TapQueue& q, float* __restrict__ gsp, bool outside, long off, float wt,
                                          const float (&dw8)[8]) {
    const unsigned long long mask = __ballot(outside);
    if (mask == 0) return;                                   // wave-uniform
    if (__popcll(mask) <= kQueueDirect) {                    // a few stragglers (smooth depth maps): not worth the detour
        if (outside) {
#pragma unroll
            for (int c = 0; c < 8; ++c) unsafeAtomicAdd(gsp + off + c, wt * dw8[c]);
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    if (outside) {
        const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        q.off[slot] = (int)off;
#pragma unroll
        for (int c = 0; c < 8; ++c) q.val[slot][c] = wt * dw8[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = __popcll(mask) * 8;
    for (int i = lane; i < n; i += 64) unsafeAtomicAdd(gsp + q.off[i >> 3] + (i & 7), q.val[i >> 3][i & 7]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                         // the records are consumed before the next tap overwrites them
}

// The four taps of one pixel for one 8-channel pass: into the window (64-bit fixed-point LDS atomics) where they fall
// inside it, queued for the coalesced global atomics otherwise.  win = &window[0][0][0] with channel pitch `cpitch` and
// row pitch `rpitch` (u64 units).
struct TapPlace {
    int ax, bx, ay, by;
    bool iax, ibx, iay, iby;
    long o00, o01, o10, o11;
};

__device__ __forceinline__ void scatter_taps(u64* win, int cpitch, int rpitch, TapQueue& q, float* __restrict__ gsp, bool valid,
                                             const mv::Taps& t, const TapPlace& w, int cbase, const float (&dw8)[8],
                                             float fxs) {
    const bool wnw = valid && t.nw != 0.0f, wne = valid && t.ne != 0.0f;
    const bool wsw = valid && t.sw != 0.0f, wse = valid && t.se != 0.0f;
    if (wnw && w.iax && w.iay) {
#pragma unroll
        for (int cl = 0; cl < 8; ++cl) fix_add(win + cl * cpitch + w.ay * rpitch + w.ax, t.nw * dw8[cl], fxs);
    }
    if (wne && w.ibx && w.iay) {
#pragma unroll
        for (int cl = 0; cl < 8; ++cl) fix_add(win + cl * cpitch + w.ay * rpitch + w.bx, t.ne * dw8[cl], fxs);
    }
    if (wsw && w.iax && w.iby) {
#pragma unroll
        for (int cl = 0; cl < 8; ++cl) fix_add(win + cl * cpitch + w.by * rpitch + w.ax, t.sw * dw8[cl], fxs);
    }
    if (wse && w.ibx && w.iby) {
#pragma unroll
        for (int cl = 0; cl < 8; ++cl) fix_add(win + cl * cpitch + w.by * rpitch + w.bx, t.se * dw8[cl], fxs);
    }
    queue_tap(q, gsp, wnw && !(w.iax && w.iay), w.o00 + cbase, t.nw, dw8);
    queue_tap(q, gsp, wne && !(w.ibx && w.iay), w.o01 + cbase, t.ne, dw8);
    queue_tap(q, gsp, wsw && !(w.iax && w.iby), w.o10 + cbase, t.sw, dw8);
    queue_tap(q, gsp, wse && !(w.ibx && w.iby), w.o11 + cbase, t.se, dw8);
}

constexpr int kWinX = 96, kWinY = 6;
static const bool g_bwd_no_tiles = MV_PROBE_ENV("MVSTER_BWD_NO_TILES") != nullptr;   // experiment switch
[[maybe_unused]] static const bool g_pix = MV_PROBE_ENV("MVSTER_PIX") != nullptr;   // experiment switch: pixel-major kernel at the fine stages

// REC: the sorted-scatter form (see above) -- no scatter window, no tap queues; pass 2 appends records instead.
template <int C, int G, bool GROUP, int DMAX, bool REC = false>
__global__ void __launch_bounds__(64 * DMAX) warp_agg_bwd_kernel(WarpAggBwdArgs ba) {
    const WarpAggArgs& a = ba.f;
    constexpr int CG = C / G;
    constexpr int NB = C / 8;                // 8-channel blocks: one scatter-window pass each
    constexpr bool GO_REG = G <= 8;          // the pixel's G output gradients live in registers (else they are re-read)
    static_assert(C % 8 == 0 && (GROUP ? (C % G == 0 && G <= 8) : (C == G)), "layout");
    __shared__ float sc[2][DMAX][64];
    __shared__ float sd[2][DMAX][64];
    __shared__ u64 gref[C][64];
    __shared__ u64 win[REC ? 1 : 8][REC ? 1 : kWinY][REC ? 1 : kWinX];
    __shared__ int worg[2][2];
    __shared__ TapQueue tapq[REC ? 1 : DMAX];          // one per wavefront
    __shared__ int lcount[REC ? kRecMaxTiles : 1];     // REC: this workgroup's samples per source tile (one view at a time)
    __shared__ int lbase[REC ? kRecMaxTiles : 1];      //      and the first list position reserved for them
    const FixScale fx = make_fix_scale(ba.maxima, G, a.D, CG, GROUP, a.fuse_d != 0, a.attn_temp);

    const mv::GridNorm gn = mv::make_grid_norm(a.Hs, a.Ws);
    const int tx = threadIdx.x;
    const int d = threadIdx.y;
    const int tid = d * 64 + tx;
    const int nthr = 64 * blockDim.y;
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const int p0 = xcd_remap(blockIdx.x, gridDim.x) * 64;
    const int p = p0 + tx;
    const bool valid = p < hw;
    const int pc = valid ? p : hw - 1;
    const int y = pc / a.w;
    const int x = pc - y * a.w;
    const long o = ((long)b * a.D + d) * hw + pc;
    const float depth = a.hypo[o];
    const float* rp = a.ref + (long)b * a.ref_bs + (long)pc * C;
    const float W = ba.wsum[o];
    const float invW = 1.0f / W;
    const float vmask = valid ? 1.0f : 0.0f;
    const float* gop = ba.grad_out + o * G;

    // go[g] = dL/d out[g] of this (pixel, d); common = sum_g go[g] * out[g]
    float go[GO_REG ? G : 1];
    float common = 0.0f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float gv = gop[g] * vmask;
        common = fmaf(gv, ba.fwd_out[o * G + g], common);
        if (GO_REG) go[g] = gv;
    }
    for (int i = tid; i < C * 64; i += nthr) (&gref[0][0])[i] = 0;
    // narrow maps: this thread's share of the reference gradient is summed over the views in registers (integers: the same
    // bits as adding every view's term to LDS) and lands in LDS once
    constexpr bool GREF_REG = C <= 16;
    u64 gacc[GREF_REG ? C : 1];
#pragma unroll
    for (int c = 0; c < (GREF_REG ? C : 1); ++c) gacc[c] = 0;
    if (!REC)
        for (int i = tid; i < 8 * kWinY * kWinX; i += nthr) (&win[0][0][0])[i] = 0;
    const int ntiles = REC ? ba.tiles_x * ba.tiles_y : 0;
    if (REC)
        for (int i = tid; i < ntiles; i += nthr) lcount[i] = 0;
    __syncthreads();

    for (int v = 0; v < a.NV; ++v) {
        mv::RT m;
        {
            const float* r = a.rt + ((long)b * a.NV + v) * 12;
#pragma unroll
            for (int i = 0; i < 9; ++i) m.r[i] = r[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) m.t[i] = r[9 + i];
        }
        float sx, sy;
        mv::project(m, (float)x, (float)y, depth, gn, sx, sy);      // (shared reciprocals: the forward kernels' form)
        mv::Taps t = mv::make_taps(sx, sy, a.Hs, a.Ws);
        const mv::TapsClamped tc = mv::clamp_taps(t, a.Hs, a.Ws);
        const long voff = (long)v * a.src_vs + (long)b * a.src_bs;
        const long o00 = ((long)tc.ya * a.Ws + tc.xa) * C, o01 = ((long)tc.ya * a.Ws + tc.xb) * C;
        const long o10 = ((long)tc.yb * a.Ws + tc.xa) * C, o11 = ((long)tc.yb * a.Ws + tc.xb) * C;
        const float* sp = a.src + voff;
        float* gsp = ba.grad_src + voff;
        // REC: the tiles this sample's weighted taps fall into, its tap mask and fractions (the record's header)
        unsigned tmask = 0;
        SampleTiles stl;
        stl.tile[0] = stl.tile[1] = stl.tile[2] = stl.tile[3] = -1;
        int li[4] = {0, 0, 0, 0};
        float wx1 = 0.0f, wy1 = 0.0f;
        if (REC) {
            tmask = valid ? tap_mask(t) : 0u;
            stl = sample_tiles(t, tmask, ba.tiles_x);
            mv::tap_fractions(sx, sy, a.Hs, a.Ws, wx1, wy1);
        }

        // pass 1: score (same arithmetic and order as the forward) and gdot = sum_g go[g] * cor[g]
        // (the warped feature of every channel stays in registers for pass 2 where the register file has room: with
        //  unrelated hypotheses on neighbouring pixels the taps are 32-byte gathers that miss L2, and the second gather
        //  round was half of this kernel's memory time)
        constexpr bool KEEP_WV = C <= 32;
        float wvk[KEEP_WV ? C : 1];
        float score = 0.0f, gdot = 0.0f, part = 0.0f;
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            f32x4 gq[2];
            if (!GO_REG) { gq[0] = ld4(gop + cb * 8) * vmask; gq[1] = ld4(gop + cb * 8 + 4) * vmask; }
#pragma unroll
            for (int c0 = 0; c0 < 8; c0 += 4) {
                const f32x4 R = ld4(rp + cb * 8 + c0);
                const f32x4 A = ld4(sp + o00 + cb * 8 + c0), Bq = ld4(sp + o01 + cb * 8 + c0);
                const f32x4 Cq = ld4(sp + o10 + cb * 8 + c0), Dq = ld4(sp + o11 + cb * 8 + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = cb * 8 + c0 + j;
                    const float wv = mv::blend(t, A[j], Bq[j], Cq[j], Dq[j]);
                    if (KEEP_WV) wvk[KEEP_WV ? c : 0] = wv;
                    if (GROUP) {
                        const float pr = mv::mul_rn(wv, R[j]);
                        part = (c % CG == 0) ? pr : mv::add_rn(part, pr);
                        if (c % CG == CG - 1) {
                            const float cg = mv::div_rn(part, (float)CG);
                            score = (c / CG == 0) ? cg : mv::add_rn(score, cg);
                            gdot = fmaf(go[GO_REG ? c / CG : 0], cg, gdot);
                        }
                    } else {
                        const float df = mv::sub_rn(R[j], wv);
                        const float cg = mv::mul_rn(df, df);
                        score = (c == 0) ? cg : mv::add_rn(score, cg);
                        gdot = fmaf(GO_REG ? go[GO_REG ? c : 0] : gq[c0 / 4][j], cg, gdot);
                    }
                }
            }
        }
        if (a.fuse_d) score = mv::div_rn(score, a.attn_temp);
        sc[v & 1][d][tx] = score;
        if (tid == 0) { worg[v & 1][0] = 0x7fffffff; worg[v & 1][1] = 0x7fffffff; }
        __syncthreads();
        if (REC) {
            // position of this sample among the workgroup's samples of each tile it touches
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
                if (stl.tile[k2] >= 0) li[k2] = atomicAdd(&lcount[stl.tile[k2]], 1);
        }
        // window origin = smallest tap coordinates of the taps that carry weight (one LDS atomic per wave)
        if (!REC) {
            const bool any = valid && (t.nw != 0.0f || t.ne != 0.0f || t.sw != 0.0f || t.se != 0.0f);
            int mnx = any ? tc.xa : 0x7fffffff, mny = any ? tc.ya : 0x7fffffff;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                mnx = min(mnx, __shfl_xor(mnx, off));
                mny = min(mny, __shfl_xor(mny, off));
            }
            if (tx == 0) { atomicMin(&worg[v & 1][0], mnx); atomicMin(&worg[v & 1][1], mny); }
        }
        float mx = sc[v & 1][0][tx];
        int dstar = 0;
        for (int j = 1; j < a.D; ++j) {
            const float sj = sc[v & 1][j][tx];
            if (sj > mx) { mx = sj; dstar = j; }          // first maximum
        }
        float den = 0.0f;
        for (int j = 0; j < a.D; ++j) den += expf(sc[v & 1][j][tx] - mx);
        const float sig = expf(score - mx) / den;
        // dL/dw = (sum_g go[g]*cor[g] - sum_g go[g]*out[g]) / W  per (pixel, d); attn_fuse_d: one weight per (pixel, d),
        // w = softmax_d(score / temp) / sqrt(C); otherwise one weight per pixel, w = max_d softmax_d(score) = 1 / den
        float wgt, dscore;
        if (a.fuse_d) {
            wgt = sig / a.sqrt_c;
            const float dsig = (gdot - common) * invW / a.sqrt_c;
            sd[v & 1][d][tx] = sig * dsig;
            __syncthreads();
            float dot = 0.0f;
            for (int j = 0; j < a.D; ++j) dot += sd[v & 1][j][tx];
            dscore = sig * (dsig - dot) / a.attn_temp;   // d/d(sum_g cor[g])
        } else {
            wgt = 1.0f / den;
            sd[v & 1][d][tx] = gdot - common;
            __syncthreads();
            float dws = 0.0f;
            for (int j = 0; j < a.D; ++j) dws += sd[v & 1][j][tx];
            dscore = dws * invW * wgt * ((d == dstar ? 1.0f : 0.0f) - sig);
        }
        long slot0[4] = {0, 0, 0, 0};
        int cstride[4] = {0, 0, 0, 0};
        if (REC) {
            // reserve list space for the workgroup's samples: ONE returning global atomic per touched tile
            const long g0 = ((long)b * a.NV + v) * ntiles;
            for (int i = tid; i < ntiles; i += nthr) {
                const int c = lcount[i];
                if (c) {
                    lbase[i] = atomicAdd(ba.rec_cursor + g0 + i, c);
                    lcount[i] = 0;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2)
                if (stl.tile[k2] >= 0) {
                    const int off = ba.rec_offset[g0 + stl.tile[k2]];
                    cstride[k2] = ba.rec_offset[g0 + stl.tile[k2] + 1] - off;
                    slot0[k2] = (long)off * NB + lbase[stl.tile[k2]] + li[k2];
                }
        }
        const int wx0 = worg[v & 1][0], wy0 = worg[v & 1][1];
        // window coordinates of the four taps (negative / too large = outside -> global atomics)
        TapPlace tp;
        tp.ax = tc.xa - wx0; tp.bx = tc.xb - wx0; tp.ay = tc.ya - wy0; tp.by = tc.yb - wy0;
        tp.iax = (unsigned)tp.ax < (unsigned)kWinX; tp.ibx = (unsigned)tp.bx < (unsigned)kWinX;
        tp.iay = (unsigned)tp.ay < (unsigned)kWinY; tp.iby = (unsigned)tp.by < (unsigned)kWinY;
        tp.o00 = o00; tp.o01 = o01; tp.o10 = o10; tp.o11 = o11;

        // pass 2: re-gather, scatter the feature gradients, 8 channels per window pass (unrolled: go[] stays in registers)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const int cbase = cb * 8;
            float dw8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dw8[j] = 0.0f;
            if (valid) {
                f32x4 gq[2];
                if (!GO_REG) { gq[0] = ld4(gop + cbase); gq[1] = ld4(gop + cbase + 4); }
#pragma unroll
                for (int c0 = 0; c0 < 8; c0 += 4) {
                    const f32x4 R = ld4(rp + cbase + c0);
                    f32x4 A = {0.f, 0.f, 0.f, 0.f}, Bq = A, Cq = A, Dq = A;
                    if (!KEEP_WV) {
                        A = ld4(sp + o00 + cbase + c0); Bq = ld4(sp + o01 + cbase + c0);
                        Cq = ld4(sp + o10 + cbase + c0); Dq = ld4(sp + o11 + cbase + c0);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cl = c0 + j;               // channel within the pass
                        const int c = cbase + cl;
                        const float gv = GO_REG ? go[GO_REG ? (GROUP ? c / CG : c) : 0] : gq[c0 / 4][j];
                        const float dcor = fmaf(gv * invW, wgt, dscore);   // direct + through the softmax
                        const float wv = KEEP_WV ? wvk[KEEP_WV ? c : 0] : mv::blend(t, A[j], Bq[j], Cq[j], Dq[j]);
                        float dref;
                        if (GROUP) {
                            dw8[cl] = dcor * (1.0f / CG) * R[j];
                            dref = dcor * (1.0f / CG) * wv;
                        } else {
                            const float df = R[j] - wv;
                            dref = 2.0f * df * dcor;
                            dw8[cl] = -dref;
                        }
                        if (GREF_REG) gacc[GREF_REG ? c : 0] += fix_cvt(dref * fx.s);
                        else fix_add(&gref[c][tx], dref, fx.s);
                    }
                }
            }
            if (REC) {
                // one 48-byte record per touched tile into the list of (batch, view, tile), channel block cb
                const float hdr = __int_as_float((int)((unsigned)(t.x0 + 1) | ((unsigned)(t.y0 + 1) << 13) | (tmask << 26)));
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2)
                    if (stl.tile[k2] >= 0) {
                        float* r = ba.rec + (slot0[k2] + (long)cb * cstride[k2]) * kRecWords;
                        st4(r, f32x4{hdr, wx1, wy1, 0.0f});
                        st4(r + 4, f32x4{dw8[0], dw8[1], dw8[2], dw8[3]});
                        st4(r + 8, f32x4{dw8[4], dw8[5], dw8[6], dw8[7]});
                    }
                continue;
            }
            scatter_taps(&win[0][0][0], kWinY * kWinX, kWinX, tapq[d], gsp, valid, t, tp, cbase, dw8, fx.s);
            __syncthreads();
            if (ba.windows) {
                // store (and clear) the window densely, texel-major / channel-fastest; scatter_gather_kernel sums the
                // windows that cover a source texel in workgroup order: no atomics, the same bits on every run
                const long slot = (((long)b * gridDim.x + blockIdx.x) * a.NV + v) * NB + cb;
                float* wp = ba.windows + slot * (8 * kWinY * kWinX);
                for (int i = tid; i < 8 * kWinY * kWinX; i += nthr) {
                    const int cl = i % 8, tex = i / 8;
                    const int wxx = tex % kWinX, wyy = tex / kWinX;
                    wp[i] = fix_get(win[cl][wyy][wxx], fx.inv);
                    win[cl][wyy][wxx] = 0;
                }
                if (cb == 0 && tid == 0) {
                    int* op = ba.win_org + (((long)b * gridDim.x + blockIdx.x) * a.NV + v) * 2;
                    op[0] = wx0; op[1] = wy0;
                }
            } else {
                // flush (and clear) the window: one global atomic per touched (texel, channel), channel-fastest: the
                // atomics of a wave go to 8 neighbouring texels x 8 channels = 256 contiguous bytes
                for (int i = tid; i < 8 * kWinY * kWinX; i += nthr) {
                    const int cl = i % 8, tex = i / 8;
                    const int wxx = tex % kWinX, wyy = tex / kWinX;
                    const u64 raw = win[cl][wyy][wxx];
                    if (raw != 0) {
                        win[cl][wyy][wxx] = 0;
                        unsafeAtomicAdd(gsp + ((long)(wy0 + wyy) * a.Ws + (wx0 + wxx)) * C + cbase + cl, fix_get(raw, fx.inv));
                    }
                }
            }
            __syncthreads();
        }
    }
    if (GREF_REG) {
#pragma unroll
        for (int c = 0; c < (GREF_REG ? C : 1); ++c) atomicAdd(&gref[c][tx], gacc[c]);
    }
    if (REC || GREF_REG) __syncthreads();      // (the window form's last flush ends with a barrier, before these adds)
    // every reference pixel of this workgroup is complete: plain coalesced stores
    float* grp = ba.grad_ref + (long)b * a.ref_bs + (long)p0 * C;
    const int npix = min(64, hw - p0);
    for (int i = tid; i < npix * C; i += nthr) grp[i] = fix_get(gref[i % C][i / C], fx.inv);
}

// 2-D tile form for the full-resolution stage (C = 8, where this kernel's time is): one workgroup owns 64 columns x R
// consecutive reference rows and keeps ONE scatter window per view for all of them -- a bilinear footprint makes
// neighbouring reference rows hit the same two source rows, so R rows need R + 2 window rows instead of 3 R.  Same
// arithmetic as warp_agg_bwd_kernel; kTileWinX columns (the 64-bit accumulators double the window's LDS footprint; 80 is
// what leaves room for two workgroups per CU).
constexpr int kTileWinX = 80;

// NW = wavefronts (depth hypotheses) the LDS is sized for.  With NW = 4 (the shipped full-resolution stage) the kernel
// needs 80.7 KB: two workgroups per CU -- it is latency-bound, and at one workgroup per CU (99 KB, when the tap queues
// were first added with an 88-wide window) the smooth-depth case ran 1.07 instead of 0.77 ms.
template <int G, bool GROUP, int R, int NW>
__global__ void __launch_bounds__(64 * NW) warp_agg_bwd_tile_kernel(WarpAggBwdArgs ba, int tiles_x) {
    const WarpAggArgs& a = ba.f;
    constexpr int C = 8;
    constexpr int CG = C / G;
    constexpr int WY = R + 6;
    __shared__ float sc[2][NW][64];
    __shared__ float sd[2][NW][64];
    __shared__ u64 gref[R][C][64];
    __shared__ u64 win[8][WY][kTileWinX];
    __shared__ int worg[2];
    __shared__ TapQueue tapq[NW];            // one per wavefront
    const FixScale fx = make_fix_scale(ba.maxima, G, a.D, CG, GROUP, true, a.attn_temp);

    const int tx = threadIdx.x;
    const int d = threadIdx.y;
    const int tid = d * 64 + tx;
    const int nthr = 64 * blockDim.y;
    const int b = blockIdx.y;
    const int hw = a.h * a.w;
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const int x0 = (int)(bid % tiles_x) * 64, y0 = (int)(bid / tiles_x) * R;
    const int x = min(x0 + tx, a.w - 1);
    const bool vcol = x0 + tx < a.w;

    for (int i = tid; i < R * C * 64; i += nthr) (&gref[0][0][0])[i] = 0;
    for (int i = tid; i < 8 * WY * kTileWinX; i += nthr) (&win[0][0][0])[i] = 0;

    for (int v = 0; v < a.NV; ++v) {
        mv::RT m;
        {
            const float* rr = a.rt + ((long)b * a.NV + v) * 12;
#pragma unroll
            for (int i = 0; i < 9; ++i) m.r[i] = rr[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) m.t[i] = rr[9 + i];
        }
        const long voff = (long)v * a.src_vs + (long)b * a.src_bs;
        const float* sp = a.src + voff;
        float* gsp = ba.grad_src + voff;
        // window origin over all rows of the tile
        if (tid == 0) { worg[0] = 0x7fffffff; worg[1] = 0x7fffffff; }
        __syncthreads();
        {
            int mnx = 0x7fffffff, mny = 0x7fffffff;
            for (int r = 0; r < R; ++r) {
                const int y = y0 + r;
                if (y >= a.h) break;
                const float depth = a.hypo[((long)b * a.D + d) * hw + (long)y * a.w + x];
                float sx, sy;
                mv::project(m, (float)x, (float)y, depth, a.Hs, a.Ws, sx, sy);
                mv::Taps t = mv::make_taps(sx, sy, a.Hs, a.Ws);
                const mv::TapsClamped tc = mv::clamp_taps(t, a.Hs, a.Ws);
                if (vcol && (t.nw != 0.0f || t.ne != 0.0f || t.sw != 0.0f || t.se != 0.0f)) {
                    mnx = min(mnx, tc.xa);
                    mny = min(mny, tc.ya);
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                mnx = min(mnx, __shfl_xor(mnx, off));
                mny = min(mny, __shfl_xor(mny, off));
            }
            if (tx == 0) { atomicMin(&worg[0], mnx); atomicMin(&worg[1], mny); }
        }
        __syncthreads();
        const int wx0 = worg[0], wy0 = worg[1];

        for (int r = 0; r < R; ++r) {
            const int y = min(y0 + r, a.h - 1);
            const bool valid = vcol && y0 + r < a.h;
            const long o = ((long)b * a.D + d) * hw + (long)y * a.w + x;
            const float depth = a.hypo[o];
            const float* rp = a.ref + (long)b * a.ref_bs + ((long)y * a.w + x) * C;
            const float invW = 1.0f / ba.wsum[o];
            float go[G];
            float common = 0.0f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                go[g] = valid ? ba.grad_out[o * G + g] : 0.0f;
                common = fmaf(go[g], ba.fwd_out[o * G + g], common);
            }
            float sx, sy;
            mv::project(m, (float)x, (float)y, depth, a.Hs, a.Ws, sx, sy);
            mv::Taps t = mv::make_taps(sx, sy, a.Hs, a.Ws);
            const mv::TapsClamped tc = mv::clamp_taps(t, a.Hs, a.Ws);
            const long o00 = ((long)tc.ya * a.Ws + tc.xa) * C, o01 = ((long)tc.ya * a.Ws + tc.xb) * C;
            const long o10 = ((long)tc.yb * a.Ws + tc.xa) * C, o11 = ((long)tc.yb * a.Ws + tc.xb) * C;

            // the eight channels of the reference pixel and of its warped source value stay in registers for both passes
            float Rv[8], wv[8];
#pragma unroll
            for (int c0 = 0; c0 < 8; c0 += 4) {
                const f32x4 Rq = ld4(rp + c0);
                const f32x4 A = ld4(sp + o00 + c0), Bq = ld4(sp + o01 + c0);
                const f32x4 Cq = ld4(sp + o10 + c0), Dq = ld4(sp + o11 + c0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Rv[c0 + j] = Rq[j];
                    wv[c0 + j] = mv::blend(t, A[j], Bq[j], Cq[j], Dq[j]);
                }
            }
            // pass 1: score (same arithmetic and order as the forward) and gdot = sum_g go[g] * cor[g]
            float score = 0.0f, gdot = 0.0f, part = 0.0f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (GROUP) {
                    const float pr = mv::mul_rn(wv[c], Rv[c]);
                    part = (c % CG == 0) ? pr : mv::add_rn(part, pr);
                    if (c % CG == CG - 1) {
                        const float cg = mv::div_rn(part, (float)CG);
                        score = (c / CG == 0) ? cg : mv::add_rn(score, cg);
                        gdot = fmaf(go[c / CG], cg, gdot);
                    }
                } else {
                    const float df = mv::sub_rn(Rv[c], wv[c]);
                    const float cg = mv::mul_rn(df, df);
                    score = (c == 0) ? cg : mv::add_rn(score, cg);
                    gdot = fmaf(go[GROUP ? 0 : c], cg, gdot);
                }
            }
            score = mv::div_rn(score, a.attn_temp);
            const int par = r & 1;
            sc[par][d][tx] = score;
            __syncthreads();
            float mx = sc[par][0][tx];
            for (int j = 1; j < a.D; ++j) mx = fmaxf(mx, sc[par][j][tx]);
            float den = 0.0f;
            for (int j = 0; j < a.D; ++j) den += expf(sc[par][j][tx] - mx);
            const float sig = expf(score - mx) / den;
            const float wgt = sig / a.sqrt_c;
            const float dsig = (gdot - common) * invW / a.sqrt_c;
            sd[par][d][tx] = sig * dsig;
            __syncthreads();
            float dot = 0.0f;
            for (int j = 0; j < a.D; ++j) dot += sd[par][j][tx];
            const float dscore = sig * (dsig - dot) / a.attn_temp;
            TapPlace tp;
            tp.ax = tc.xa - wx0; tp.bx = tc.xb - wx0; tp.ay = tc.ya - wy0; tp.by = tc.yb - wy0;
            tp.iax = (unsigned)tp.ax < (unsigned)kTileWinX; tp.ibx = (unsigned)tp.bx < (unsigned)kTileWinX;
            tp.iay = (unsigned)tp.ay < (unsigned)WY; tp.iby = (unsigned)tp.by < (unsigned)WY;
            tp.o00 = o00; tp.o01 = o01; tp.o10 = o10; tp.o11 = o11;

            // pass 2: scatter into the tile's window (64-bit fixed-point LDS atomics; coalesced global atomics outside it)
            float dw8[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int g = GROUP ? c / CG : c;
                const float dcor = fmaf(go[g] * invW, wgt, dscore);
                float dref;
                if (GROUP) {
                    dw8[c] = dcor * (1.0f / CG) * Rv[c];
                    dref = dcor * (1.0f / CG) * wv[c];
                } else {
                    const float df = Rv[c] - wv[c];
                    dref = 2.0f * df * dcor;
                    dw8[c] = -dref;
                }
                if (valid) fix_add(&gref[r][c][tx], dref, fx.s);
            }
            scatter_taps(&win[0][0][0], WY * kTileWinX, kTileWinX, tapq[d], gsp, valid, t, tp, 0, dw8, fx.s);
        }
        __syncthreads();
        if (ba.windows) {
            // dense, texel-major / channel-fastest window for scatter_gather_kernel (fixed summation order)
            const long slot = ((long)b * gridDim.x + blockIdx.x) * a.NV + v;
            float* wp = ba.windows + slot * (8 * WY * kTileWinX);
            for (int i = tid; i < 8 * WY * kTileWinX; i += nthr) {
                const int cl = i % 8, tex = i / 8;
                const int wxx = tex % kTileWinX, wyy = tex / kTileWinX;
                wp[i] = fix_get(win[cl][wyy][wxx], fx.inv);
                win[cl][wyy][wxx] = 0;
            }
            if (tid == 0) { ba.win_org[slot * 2] = wx0; ba.win_org[slot * 2 + 1] = wy0; }
        } else {
            for (int i = tid; i < 8 * WY * kTileWinX; i += nthr) {
                const int cl = i % 8, tex = i / 8;
                const int wxx = tex % kTileWinX, wyy = tex / kTileWinX;
                const u64 raw = win[cl][wyy][wxx];
                if (raw != 0) {
                    win[cl][wyy][wxx] = 0;
                    unsafeAtomicAdd(gsp + ((long)(wy0 + wyy) * a.Ws + (wx0 + wxx)) * C + cl, fix_get(raw, fx.inv));
                }
            }
        }
        __syncthreads();
    }
    // reference gradients of the tile: plain stores
    for (int r = 0; r < R; ++r) {
        const int y = y0 + r;
        if (y >= a.h) break;
        float* grp = ba.grad_ref + (long)b * a.ref_bs + ((long)y * a.w + x0) * C;
        const int npix = min(64, a.w - x0);
        for (int i = tid; i < npix * C; i += nthr) grp[i] = fix_get(gref[r][i % C][i / C], fx.inv);
    }
}

// Second pass of the atomic-free backward: one workgroup per 32 x 4 tile of one source map; lane = (texel, channel
// quad).  The windows that overlap the tile are found by testing every workgroup's origin (a few thousand integer
// compares per tile, in index order through a ballot prefix so that the summation order is fixed) and summed in
// registers; the tile is then added onto grad_src, which holds what the first pass scattered directly (taps outside a
// window).  WY = window rows (kWinY for the row kernel, R + 6 for the tile kernel).
constexpr int kGatherList = 256;

template <int WY, int WX, int NBLK>
__global__ void __launch_bounds__(256) scatter_gather_kernel(const float* __restrict__ windows, const int* __restrict__ org,
                                                             float* __restrict__ grad_src, int nblk, int NV, int Hs, int Ws,
                                                             long src_vs, long src_bs, int tiles_x) {
    constexpr int C = NBLK * 8;
    __shared__ int list[kGatherList];
    __shared__ int wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = blockIdx.y, b = blockIdx.z;
    const int tx0 = (int)(blockIdx.x % tiles_x) * 32, ty0 = (int)(blockIdx.x / tiles_x) * 4;
    const int half = tid & 1, tex = tid >> 1;
    const int tx = tx0 + (tex & 31), ty = ty0 + (tex >> 5);
    const bool inside = tx < Ws && ty < Hs;
    f32x4 acc[NBLK];
#pragma unroll
    for (int cb = 0; cb < NBLK; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long slot0 = (long)b * nblk;

    int count = 0;                                   // entries in list (block-uniform)
    for (int base = 0; base < nblk || count > 0; base += 256) {
        if (base < nblk) {
            const int k = base + tid;
            bool hit = false;
            if (k < nblk) {
                const int* o = org + ((slot0 + k) * NV + v) * 2;
                const int wx0 = o[0], wy0 = o[1];
                hit = wx0 != 0x7fffffff && wx0 < tx0 + 32 && wx0 + WX > tx0 && wy0 < ty0 + 4 && wy0 + WY > ty0;
            }
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcnt[wave] = __popcll(m);
            __syncthreads();
            int off = count + __popcll(m & ((1ull << lane) - 1));
            for (int w = 0; w < wave; ++w) off += wcnt[w];
            if (hit) list[off] = k;                  // (count + 256 <= kGatherList is kept by draining below)
            count += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        // drain when the next chunk might not fit, and at the end
        if (count > kGatherList - 256 || base + 256 >= nblk) {
            for (int li = 0; li < count; ++li) {
                const int k = list[li];
                const int* o = org + ((slot0 + k) * NV + v) * 2;
                const int lx = tx - o[0], ly = ty - o[1];
                if ((unsigned)lx < (unsigned)WX && (unsigned)ly < (unsigned)WY) {
                    const float* wp = windows + ((slot0 + k) * NV + v) * ((long)NBLK * 8 * WY * WX) +
                                      ((long)ly * WX + lx) * 8 + half * 4;
#pragma unroll
                    for (int cb = 0; cb < NBLK; ++cb) acc[cb] += ld4(wp + (long)cb * (8 * WY * WX));
                }
            }
            __syncthreads();
            count = 0;
            if (base + 256 >= nblk) break;
        }
    }
    if (inside) {
        float* dst = grad_src + (long)v * src_vs + (long)b * src_bs + ((long)ty * Ws + tx) * C + half * 4;
#pragma unroll
        for (int cb = 0; cb < NBLK; ++cb) st4(dst + cb * 8, ld4(dst + cb * 8) + acc[cb]);
    }
}

template <int WY, int WX, int NBLK>
int launch_gather(const WarpAggBwdArgs& ba, int nblk, hipStream_t stream) {
    const WarpAggArgs& a = ba.f;
    const int tiles_x = (a.Ws + 31) / 32, tiles_y = (a.Hs + 3) / 4;
    hipLaunchKernelGGL((scatter_gather_kernel<WY, WX, NBLK>), dim3(tiles_x * tiles_y, a.NV, a.B), dim3(256), 0, stream, ba.windows,
                       ba.win_org, ba.grad_src, nblk, a.NV, a.Hs, a.Ws, a.src_vs, a.src_bs, tiles_x);
    return mv_check_launch();
}

// ---- sorted scatter: K0 (count), scan, K2 (accumulate) ---------------------------------------------------------------
// K0: thread = (pixel, hypothesis) of batch item blockIdx.y, all views; per view an LDS histogram over the source tiles,
// flushed with one global atomic per touched tile.
__device__ __forceinline__ void warp_bwd_count_block(const WarpAggArgs& a, int* __restrict__ count, int tiles_x, int tiles_y,
                                                     int bx, int b) {
    __shared__ int lcount[kRecMaxTiles];
    const int ntiles = tiles_x * tiles_y;
    const long hw = (long)a.h * a.w;
    const long i = (long)bx * 256 + threadIdx.x;                    // = pixel * D + d
    const mv::GridNorm gn = mv::make_grid_norm(a.Hs, a.Ws);
    const bool valid = i < hw * a.D;
    const long pc = valid ? i / a.D : hw - 1;
    const int d = valid ? (int)(i - pc * a.D) : 0;
    const int y = (int)(pc / a.w), x = (int)(pc - (long)y * a.w);
    const float depth = a.hypo[((long)b * a.D + d) * hw + pc];
    for (int t = threadIdx.x; t < ntiles; t += 256) lcount[t] = 0;
    __syncthreads();
    for (int v = 0; v < a.NV; ++v) {
        mv::RT m;
        const float* r = a.rt + ((long)b * a.NV + v) * 12;
#pragma unroll
        for (int k = 0; k < 9; ++k) m.r[k] = r[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) m.t[k] = r[9 + k];
        float sx, sy;
        mv::project(m, (float)x, (float)y, depth, gn, sx, sy);      // (shared reciprocals: the forward kernels' form)
        mv::Taps t = mv::make_taps(sx, sy, a.Hs, a.Ws);
        mv::clamp_taps(t, a.Hs, a.Ws);
        const SampleTiles stl = sample_tiles(t, valid ? tap_mask(t) : 0u, tiles_x);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (stl.tile[k] >= 0) atomicAdd(&lcount[stl.tile[k]], 1);
        __syncthreads();
        int* gc = count + ((long)b * a.NV + v) * ntiles;
        for (int tl = threadIdx.x; tl < ntiles; tl += 256) {
            const int c = lcount[tl];
            if (c) {
                atomicAdd(gc + tl, c);
                lcount[tl] = 0;
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) warp_bwd_zero_kernel(int* __restrict__ a, long na, int* __restrict__ b, long nb) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < na) a[i] = 0;
    if (i < nb) b[i] = 0;
}

__global__ void __launch_bounds__(256) warp_bwd_count_kernel(WarpAggArgs a, int* __restrict__ count, int tiles_x, int tiles_y) {
    warp_bwd_count_block(a, count, tiles_x, tiles_y, blockIdx.x, blockIdx.y);
}

// The counting pass and the operand maxima of the fixed-point scale in ONE launch: they are independent, one is bound by
// instructions and the other by memory, and each alone leaves most of the chip idle for part of its time.  Workgroups
// [0, nabs) reduce the maxima, the rest count (per_b workgroups per batch item).
__global__ void __launch_bounds__(256) warp_bwd_prep_kernel(WarpAggArgs a, int* __restrict__ count, int tiles_x, int tiles_y,
                                                            AbsMaxArgs am, float* __restrict__ maxima, int nabs, int per_b) {
    if ((int)blockIdx.x < nabs) {
        absmax_block(am, maxima, blockIdx.x);
        return;
    }
    const int r = (int)blockIdx.x - nabs;
    warp_bwd_count_block(a, count, tiles_x, tiles_y, r % per_b, r / per_b);
}

// offset[0..n] = exclusive prefix sums of count[0..n) (offset[n] = total); one workgroup
__global__ void __launch_bounds__(1024) warp_bwd_scan_kernel(const int* __restrict__ count, int* __restrict__ offset, int n) {
    __shared__ int part[1024];
    const int chunk = (n + 1023) / 1024;
    const int lo = threadIdx.x * chunk, hi = min(lo + chunk, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;                                 // exclusive
    for (int i = lo; i < hi; ++i) {
        offset[i] = run;
        run += count[i];
    }
    if (threadIdx.x == 1023) offset[n] = part[1023];
}

struct WarpBwdAccumArgs {
    const float* rec; const int* offset; const float* maxima; float* grad_src;
    long src_vs, src_bs;
    int NV, NB, C, Hs, Ws, tiles_x, tiles_y, G, D, CG, group, fuse;
    float attn_temp;
};

// K2: workgroup = (tile, view * NB + channel block, batch); lane = record.  The window is channel-major ([8] planes of
// 32 x 32 u64): the lanes of an atomic instruction hit one plane at (practically) random texels.
__global__ void __launch_bounds__(256) warp_bwd_accum_kernel(WarpBwdAccumArgs a) {
    __shared__ u64 win[8][kRecTileX * kRecTileY];
    const FixScale fx = make_fix_scale(a.maxima, a.G, a.D, a.CG, a.group != 0, a.fuse != 0, a.attn_temp);
    const int tile = blockIdx.x, v = blockIdx.y / a.NB, cb = blockIdx.y - v * a.NB, b = blockIdx.z;
    const int ntiles = a.tiles_x * a.tiles_y;
    const int tile_y = tile / a.tiles_x, tile_x = tile - tile_y * a.tiles_x;
    for (int i = threadIdx.x; i < 8 * kRecTileX * kRecTileY; i += 256) (&win[0][0])[i] = 0;
    __syncthreads();
    const long g0 = ((long)b * a.NV + v) * ntiles + tile;
    const int off = a.offset[g0], cnt = a.offset[g0 + 1] - off;
    const float* base = a.rec + ((long)off * a.NB + (long)cb * cnt) * kRecWords;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const float* r = base + (long)i * kRecWords;
        const f32x4 h = ld4(r), d0 = ld4(r + 4), d1 = ld4(r + 8);
        const unsigned pk = (unsigned)__float_as_int(h[0]);
        const int x0 = (int)(pk & 0x1fffu) - 1, y0 = (int)((pk >> 13) & 0x1fffu) - 1;
        const unsigned mask = pk >> 26;
        float wt[4];
        mv::tap_weights(h[1], h[2], wt[0], wt[1], wt[2], wt[3]);
        const float dw[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tx = x0 + (k & 1), ty = y0 + (k >> 1);
            if (((mask >> k) & 1u) && (tx >> kRecShiftX) == tile_x && (ty >> kRecShiftY) == tile_y) {
                const int idx = (ty & (kRecTileY - 1)) * kRecTileX + (tx & (kRecTileX - 1));
#pragma unroll
                for (int c = 0; c < 8; ++c) fix_add(&win[c][idx], wt[k] * dw[c], fx.s);
            }
        }
    }
    __syncthreads();
    // the tile's texels, every one exactly once: plain stores, channel-fastest (32 contiguous bytes per texel and block)
    float* gsp = a.grad_src + (long)v * a.src_vs + (long)b * a.src_bs;
    for (int i = threadIdx.x; i < 8 * kRecTileX * kRecTileY; i += 256) {
        const int c = i & 7, tex = i >> 3;
        const int ty = tile_y * kRecTileY + (tex >> kRecShiftX), tx = tile_x * kRecTileX + (tex & (kRecTileX - 1));
        if (ty < a.Hs && tx < a.Ws) gsp[((long)ty * a.Ws + tx) * a.C + cb * 8 + c] = fix_get(win[c][tex], fx.inv);
    }
}

constexpr int kTileR = 4;     // reference rows per workgroup of the tile kernel

// which first-pass kernel a configuration takes, its workgroups per batch item and its window height
static bool bwd_uses_tiles(int C, int G, int D, int fuse_d) {
    return C == 8 && C / G <= 8 && !g_bwd_no_tiles && fuse_d && D <= 8;
}
static int bwd_blocks(int C, int G, int D, int fuse_d, int h, int w) {
    return bwd_uses_tiles(C, G, D, fuse_d) ? ((w + 63) / 64) * ((h + kTileR - 1) / kTileR) : (h * w + 63) / 64;
}

template <int C, int G, bool GROUP>
int launch_bwd(const WarpAggBwdArgs& ba, hipStream_t stream) {
    const WarpAggArgs& a = ba.f;
    const int nblk = bwd_blocks(C, G, a.D, a.fuse_d, a.h, a.w);
    if constexpr (C == 8 && C / G <= 8) {
        if (bwd_uses_tiles(C, G, a.D, a.fuse_d)) {      // the shipped full-resolution stage
            const int tiles_x = (a.w + 63) / 64;
            if (a.D <= 4) {
                MV_NOTE_KERNEL("warp_agg_bwd_tile_kernel<%d, %s, %d, 4>", G, GROUP ? "true" : "false", kTileR);
                hipLaunchKernelGGL((warp_agg_bwd_tile_kernel<G, GROUP, kTileR, 4>), dim3(nblk, a.B), dim3(64, a.D), 0, stream,
                                   ba, tiles_x);
            } else {
                MV_NOTE_KERNEL("warp_agg_bwd_tile_kernel<%d, %s, %d, 8>", G, GROUP ? "true" : "false", kTileR);
                hipLaunchKernelGGL((warp_agg_bwd_tile_kernel<G, GROUP, kTileR, 8>), dim3(nblk, a.B), dim3(64, a.D), 0, stream,
                                   ba, tiles_x);
            }
            if (int rc = mv_check_launch()) return rc;
            return ba.windows ? launch_gather<kTileR + 6, kTileWinX, 1>(ba, nblk, stream) : MVSTER_OK;
        }
    }
    dim3 block(64, a.D);
    dim3 grid(nblk, a.B);
    if (a.D <= 8) {
        MV_NOTE_KERNEL("warp_agg_bwd_kernel<%d, %d, %s, 8>", C, G, GROUP ? "true" : "false");
        hipLaunchKernelGGL((warp_agg_bwd_kernel<C, G, GROUP, 8>), grid, block, 0, stream, ba);
    } else {
        MV_NOTE_KERNEL("warp_agg_bwd_kernel<%d, %d, %s, %d>", C, G, GROUP ? "true" : "false", kMaxD);
        hipLaunchKernelGGL((warp_agg_bwd_kernel<C, G, GROUP, kMaxD>), grid, block, 0, stream, ba);
    }
    if (int rc = mv_check_launch()) return rc;
    return ba.windows ? launch_gather<kWinY, kWinX, C / 8>(ba, nblk, stream) : MVSTER_OK;
}

// K1 of the sorted scatter: the row kernel's REC instance (all channel counts; the 2-D tile kernel's advantage was its
// shared scatter window, which this form does not have)
template <int C, int G, bool GROUP>
int launch_bwd_rec(const WarpAggBwdArgs& ba, hipStream_t stream) {
    const WarpAggArgs& a = ba.f;
    dim3 block(64, a.D);
    dim3 grid((a.h * a.w + 63) / 64, a.B);
    if (a.D <= 8) {
        MV_NOTE_KERNEL("warp_agg_bwd_kernel<%d, %d, %s, 8, true>", C, G, GROUP ? "true" : "false");
        hipLaunchKernelGGL((warp_agg_bwd_kernel<C, G, GROUP, 8, true>), grid, block, 0, stream, ba);
    } else {
        MV_NOTE_KERNEL("warp_agg_bwd_kernel<%d, %d, %s, %d, true>", C, G, GROUP ? "true" : "false", kMaxD);
        hipLaunchKernelGGL((warp_agg_bwd_kernel<C, G, GROUP, kMaxD, true>), grid, block, 0, stream, ba);
    }
    return mv_check_launch();
}

}  // namespace

// mvster_warp_agg_fwd with the stage's hypothesis scheduling fused in (wave-local kernel only: group correlation, D in
// {4, 8}, C / 8 * D <= 64 -- every stage of the shipped cascade; MVSTER_ERR_UNSUPPORTED otherwise and the caller runs the
// scheduler launch + mvster_warp_agg_fwd).  mode 1: schedule_inverse_range of inv_min / inv_max [B, h/2, w/2]
// (models/mvs4net_utils.py:79-86); mode 2: init_inverse_range of depth_values [B, ndv] (:71-77).  hypo_out [B, D, h, w]
// receives the hypotheses (bit-identical to the scheduler kernels').
extern "C" int mvster_warp_agg_fwd_sched(const float* ref_feat, const float* src_feat, const float* rt, const float* inv_min,
                                         const float* inv_max, const float* depth_values, int ndv, float* hypo_out, float* out,
                                         float* wsum_out, int B, int NV, int C, int G, int D, int h, int w, int Hs, int Ws,
                                         long ref_batch_stride, long src_view_stride, long src_batch_stride, int attn_fuse_d,
                                         float attn_temp, int mode, void* stream) {
    if (!ref_feat || !src_feat || !rt || !hypo_out || !out) return MVSTER_ERR_NULL;
    if (mode == 1 ? (!inv_min || !inv_max) : (mode == 2 ? !depth_values : true)) return mode == 1 || mode == 2 ? MVSTER_ERR_NULL : MVSTER_ERR_SHAPE;
    if (B <= 0 || NV <= 0 || h <= 0 || w <= 0 || Hs <= 0 || Ws <= 0 || (mode == 1 && ((h | w) & 1)) || (mode == 2 && ndv < 2))
        return MVSTER_ERR_SHAPE;
    if (D != 4 && D != 8) return MVSTER_ERR_UNSUPPORTED;
    WarpAggArgs a;
    a.ref = ref_feat; a.src = src_feat; a.rt = rt; a.hypo = nullptr; a.out = out; a.wsum_out = wsum_out;
    a.ref_bs = ref_batch_stride; a.src_vs = src_view_stride; a.src_bs = src_batch_stride;
    a.B = B; a.NV = NV; a.D = D; a.h = h; a.w = w; a.Hs = Hs; a.Ws = Ws;
    a.attn_temp = attn_temp; a.sqrt_c = sqrtf((float)C); a.fuse_d = attn_fuse_d;
    a.inv_min = inv_min; a.inv_max = inv_max; a.dvals = depth_values; a.hypo_out = hypo_out; a.ndv = ndv;
    hipStream_t s = (hipStream_t)stream;
    if (C == 8 && G == 4) return dispatch_fwd_wave<8, 4>(a, s, mode);
    if (C == 16 && G == 4) return dispatch_fwd_wave<16, 4>(a, s, mode);
    if (C == 32 && G == 8) return dispatch_fwd_wave<32, 8>(a, s, mode);
    if (C == 64 && G == 8) return dispatch_fwd_wave<64, 8>(a, s, mode);
    return MVSTER_ERR_UNSUPPORTED;
}

extern "C" int mvster_warp_agg_fwd(const float* ref_feat, const float* src_feat, const float* rt, const float* hypo,
                                   float* out, float* wsum_out, int B, int NV, int C, int G, int D, int h, int w,
                                   int Hs, int Ws, long ref_batch_stride, long src_view_stride, long src_batch_stride,
                                   int group_cor, int attn_fuse_d, float attn_temp, int variant, void* stream) {
    if (!ref_feat || !src_feat || !rt || !hypo || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || NV <= 0 || D <= 0 || D > kMaxFwdD || h <= 0 || w <= 0 || Hs <= 0 || Ws <= 0) return MVSTER_ERR_SHAPE;
    if (!group_cor && G != C) return MVSTER_ERR_SHAPE;
    WarpAggArgs a;
    a.ref = ref_feat; a.src = src_feat; a.rt = rt; a.hypo = hypo; a.out = out; a.wsum_out = wsum_out;
    a.ref_bs = ref_batch_stride; a.src_vs = src_view_stride; a.src_bs = src_batch_stride;
    a.B = B; a.NV = NV; a.D = D; a.h = h; a.w = w; a.Hs = Hs; a.Ws = Ws;
    a.attn_temp = attn_temp; a.sqrt_c = sqrtf((float)C); a.fuse_d = attn_fuse_d;
    a.inv_min = a.inv_max = a.dvals = nullptr; a.hypo_out = nullptr; a.ndv = 0;
    hipStream_t s = (hipStream_t)stream;
    // variant: 0 = choose; 1 = one thread per (pixel, d); 2 = workgroup-level lane split (C >= 16);
    // 3 = wave-local kernel (what 0 picks whenever it applies); 4 = pixel-major kernel (faster on cache-resident inputs,
    // slower inside the forward: kept as a tested alternative)
#ifdef MVSTER_PROBES
    if (group_cor && (D == 4 || D == 8) && (variant == 4 || (variant == 0 && C <= 16 && g_pix))) {
        int rc = MVSTER_ERR_UNSUPPORTED;
        if (C == 8 && G == 4) rc = dispatch_fwd_pix<8, 4>(a, s);
        else if (C == 8 && G == 8) rc = dispatch_fwd_pix<8, 8>(a, s);
        else if (C == 16 && G == 4) rc = dispatch_fwd_pix<16, 4>(a, s);
        else if (C == 16 && G == 8) rc = dispatch_fwd_pix<16, 8>(a, s);
        else if (C == 32 && G == 8 && variant == 4) rc = dispatch_fwd_pix<32, 8>(a, s);
        else if (C == 32 && G == 4 && variant == 4) rc = dispatch_fwd_pix<32, 4>(a, s);
        if (rc != MVSTER_ERR_UNSUPPORTED && !(rc == MVSTER_ERR_SHAPE && variant == 0)) return rc;
        // (a map too large for the 24-bit index arithmetic falls through to the wave-local kernel)
    }
    if (group_cor && (D == 4 || D == 8) && variant == 5) {      // LDS-staged source windows (the two fine stages)
        if (C == 8 && G == 4) return dispatch_fwd_tile<8, 4>(a, s);
        if (C == 8 && G == 8) return dispatch_fwd_tile<8, 8>(a, s);
        if (C == 16 && G == 4) return dispatch_fwd_tile<16, 4>(a, s);
        if (C == 16 && G == 8) return dispatch_fwd_tile<16, 8>(a, s);
        return MVSTER_ERR_UNSUPPORTED;
    }
#else
    if (variant == 4 || variant == 5) return MVSTER_ERR_UNSUPPORTED;      // (the probe library has them)
#endif
    if (group_cor && (D == 4 || D == 8) && (variant == 0 || variant == 3)) {
        if (C == 8 && G == 4) return dispatch_fwd_wave<8, 4>(a, s);
        if (C == 8 && G == 8) return dispatch_fwd_wave<8, 8>(a, s);
        if (C == 16 && G == 4) return dispatch_fwd_wave<16, 4>(a, s);
        if (C == 16 && G == 8) return dispatch_fwd_wave<16, 8>(a, s);
        if (C == 32 && G == 8) return dispatch_fwd_wave<32, 8>(a, s);
        if (C == 32 && G == 4) return dispatch_fwd_wave<32, 4>(a, s);
        if (C == 64 && G == 8) return dispatch_fwd_wave<64, 8>(a, s);
    }
    if (group_cor && D <= 8 && variant != 1) {
        if (C == 64 && G == 8) return launch_fwd_lanes<64, 8>(a, s);
        if (C == 32 && G == 8) return launch_fwd_lanes<32, 8>(a, s);
        if (C == 16 && G == 4) return launch_fwd_lanes<16, 4>(a, s);    // 47 -> 33 us at 256x320 (measured)
        if (C == 16 && G == 8) return launch_fwd_lanes<16, 8>(a, s);
    }
#define MV_CASE(CC, GG, GR) \
    if (C == CC && G == GG && (group_cor != 0) == GR) return launch_fwd<CC, GG, GR>(a, s);
    MV_CASE(64, 8, true)
    MV_CASE(32, 8, true)
    MV_CASE(16, 4, true)
    MV_CASE(8, 4, true)
    MV_CASE(16, 8, true)
    MV_CASE(8, 8, true)
    MV_CASE(64, 4, true)
    MV_CASE(32, 4, true)
    MV_CASE(8, 8, false)
    MV_CASE(16, 16, false)
    MV_CASE(32, 32, false)
    MV_CASE(64, 64, false)
#undef MV_CASE
    return MVSTER_ERR_UNSUPPORTED;
}

extern "C" int mvster_warp_agg_bwd_scratch(int B, int NV, int C, int G, int D, int h, int w, int attn_fuse_d,
                                          long* window_floats, long* origin_ints);   // (defined below, used by the launcher)

extern "C" int mvster_warp_agg_bwd_scratch(int B, int NV, int C, int G, int D, int h, int w, int attn_fuse_d,
                                          long* window_floats, long* origin_ints) {
    if (!window_floats || !origin_ints) return MVSTER_ERR_NULL;
    if (B <= 0 || NV <= 0 || C <= 0 || C % 8 || G <= 0 || D <= 0 || h <= 0 || w <= 0) return MVSTER_ERR_SHAPE;
    const long nblk = bwd_blocks(C, G, D, attn_fuse_d, h, w);
    const bool tiles = bwd_uses_tiles(C, G, D, attn_fuse_d);
    *window_floats = (long)B * nblk * NV * (C / 8) * 8 * (tiles ? (kTileR + 6) * kTileWinX : kWinY * kWinX);
    *origin_ints = (long)B * nblk * NV * 2 + 4;      // + the three operand maxima of the fixed-point scale
    return MVSTER_OK;
}

extern "C" int mvster_warp_agg_bwd(const float* ref_feat, const float* src_feat, const float* rt, const float* hypo,
                                   const float* out, const float* wsum, const float* grad_out, float* grad_ref,
                                   float* grad_src, float* windows, int* win_org, int B, int NV, int C, int G, int D, int h,
                                   int w, int Hs, int Ws, long ref_batch_stride, long src_view_stride,
                                   long src_batch_stride, int group_cor, int attn_fuse_d, float attn_temp, void* stream) {
    if (!ref_feat || !src_feat || !rt || !hypo || !out || !wsum || !grad_out || !grad_ref || !grad_src)
        return MVSTER_ERR_NULL;
    if (!win_org) return MVSTER_ERR_NULL;
    if (B <= 0 || NV <= 0 || D <= 0 || D > kMaxD || h <= 0 || w <= 0 || Hs <= 0 || Ws <= 0) return MVSTER_ERR_SHAPE;
    if (!group_cor && G != C) return MVSTER_ERR_SHAPE;
    WarpAggBwdArgs ba;
    WarpAggArgs& a = ba.f;
    a.ref = ref_feat; a.src = src_feat; a.rt = rt; a.hypo = hypo; a.out = nullptr; a.wsum_out = nullptr;
    a.ref_bs = ref_batch_stride; a.src_vs = src_view_stride; a.src_bs = src_batch_stride;
    a.B = B; a.NV = NV; a.D = D; a.h = h; a.w = w; a.Hs = Hs; a.Ws = Ws;
    a.attn_temp = attn_temp; a.sqrt_c = sqrtf((float)C); a.fuse_d = attn_fuse_d;
    ba.fwd_out = out; ba.wsum = wsum; ba.grad_out = grad_out; ba.grad_ref = grad_ref; ba.grad_src = grad_src;
    ba.windows = windows; ba.win_org = win_org;
    hipStream_t s = (hipStream_t)stream;
    {
        // operand maxima for the fixed-point scale of the LDS accumulators: the last 4 ints of win_org
        long nf = 0, ni = 0;
        mvster_warp_agg_bwd_scratch(B, NV, C, G, D, h, w, attn_fuse_d, &nf, &ni);
        float* mx = reinterpret_cast<float*>(win_org + (ni - 4));
        // (zeroed by a kernel, not a memset node: see mvster_warp_agg_bwd_sorted)
        hipLaunchKernelGGL(warp_bwd_zero_kernel, dim3(1), dim3(256), 0, s, reinterpret_cast<int*>(mx), 4L, reinterpret_cast<int*>(mx), 0L);
        const long n_go = (long)B * D * h * w * G, n_ref = (long)h * w * C, n_src = (long)Hs * Ws * C;
        if (ref_batch_stride == n_ref && src_batch_stride == n_src && src_view_stride == n_src * B) {
            const float* xs[3] = {grad_out, ref_feat, src_feat};
            const long ns[3] = {n_go, n_ref * B, n_src * B * NV};
            launch_absmax(xs, ns, 3, mx, s);
        } else {
            // strided batches: one launch per contiguous piece (each reduces into its operand's slot)
            launch_absmax(&grad_out, &n_go, 1, mx, s);
            for (int bb = 0; bb < B; ++bb) {
                const float* p = ref_feat + (long)bb * ref_batch_stride;
                launch_absmax(&p, &n_ref, 1, mx + 1, s);
            }
            for (int v = 0; v < NV; ++v)
                for (int bb = 0; bb < B; ++bb) {
                    const float* p = src_feat + (long)v * src_view_stride + (long)bb * src_batch_stride;
                    launch_absmax(&p, &n_src, 1, mx + 2, s);
                }
        }
        ba.maxima = mx;
    }
#define MV_CASE(CC, GG, GR) \
    if (C == CC && G == GG && (group_cor != 0) == GR) return launch_bwd<CC, GG, GR>(ba, s);
    MV_CASE(64, 8, true)
    MV_CASE(32, 8, true)
    MV_CASE(16, 4, true)
    MV_CASE(8, 4, true)
    MV_CASE(16, 8, true)
    MV_CASE(8, 8, true)
    MV_CASE(64, 4, true)
    MV_CASE(32, 4, true)
    MV_CASE(8, 8, false)
    MV_CASE(16, 16, false)
    MV_CASE(32, 32, false)
    MV_CASE(64, 64, false)
#undef MV_CASE
    return MVSTER_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------------------
// mvster_warp_agg_bwd with the source-view gradient accumulated by the SORTED SCATTER (see the section above the kernels):
// no global atomics, bit-reproducible, independent of the smoothness of the depth maps, and grad_src needs no zero fill
// (every texel is written exactly once).  Scratch: rec = *rec_floats floats (the worst case: every sample on a tile corner,
// 4 records per sample and 8-channel block; typically 1.07 are used), ints = *ints ints (tile counts, cursors, offsets and
// the operand maxima of the fixed-point scale).  MVSTER_ERR_UNSUPPORTED for source maps of more than 2048 tiles of 32 x 32
// texels or beyond 8190 texels a side: the caller then takes mvster_warp_agg_bwd.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int mvster_warp_agg_bwd_sorted_scratch(int B, int NV, int C, int D, int h, int w, int Hs, int Ws, long* rec_floats,
                                                 long* ints) {
    if (!rec_floats || !ints) return MVSTER_ERR_NULL;
    if (B <= 0 || NV <= 0 || C <= 0 || C % 8 || D <= 0 || h <= 0 || w <= 0 || Hs <= 0 || Ws <= 0) return MVSTER_ERR_SHAPE;
    const long tiles_x = (Ws + kRecTileX - 1) / kRecTileX, tiles_y = (Hs + kRecTileY - 1) / kRecTileY;
    if (tiles_x * tiles_y > kRecMaxTiles || Hs > 8190 || Ws > 8190) return MVSTER_ERR_UNSUPPORTED;
    const long n = (long)B * NV * tiles_x * tiles_y;
    const long samples = (long)B * h * w * D * NV;
    if (samples * 4 >= (1L << 31)) return MVSTER_ERR_UNSUPPORTED;          // (list positions are 32-bit)
    *rec_floats = samples * 4 * (C / 8) * kRecWords;
    *ints = 3 * n + 1 + 4;
    return MVSTER_OK;
}

extern "C" int mvster_warp_agg_bwd_sorted(const float* ref_feat, const float* src_feat, const float* rt, const float* hypo,
                                          const float* out, const float* wsum, const float* grad_out, float* grad_ref,
                                          float* grad_src, float* rec, int* ints, int B, int NV, int C, int G, int D, int h,
                                          int w, int Hs, int Ws, long ref_batch_stride, long src_view_stride,
                                          long src_batch_stride, int group_cor, int attn_fuse_d, float attn_temp, void* stream) {
    if (!ref_feat || !src_feat || !rt || !hypo || !out || !wsum || !grad_out || !grad_ref || !grad_src || !rec || !ints)
        return MVSTER_ERR_NULL;
    if (B <= 0 || NV <= 0 || D <= 0 || D > kMaxD || h <= 0 || w <= 0 || Hs <= 0 || Ws <= 0) return MVSTER_ERR_SHAPE;
    if (!group_cor && G != C) return MVSTER_ERR_SHAPE;
    long nf = 0, ni = 0;
    if (int rc = mvster_warp_agg_bwd_sorted_scratch(B, NV, C, D, h, w, Hs, Ws, &nf, &ni)) return rc;
    const int tiles_x = (Ws + kRecTileX - 1) / kRecTileX, tiles_y = (Hs + kRecTileY - 1) / kRecTileY;
    const long n = (long)B * NV * tiles_x * tiles_y;
    int* count = ints;
    int* cursor = ints + n;
    int* offset = ints + 2 * n;
    float* mx = reinterpret_cast<float*>(ints + 3 * n + 1);
    WarpAggBwdArgs ba;
    WarpAggArgs& a = ba.f;
    a.ref = ref_feat; a.src = src_feat; a.rt = rt; a.hypo = hypo; a.out = nullptr; a.wsum_out = nullptr;
    a.ref_bs = ref_batch_stride; a.src_vs = src_view_stride; a.src_bs = src_batch_stride;
    a.B = B; a.NV = NV; a.D = D; a.h = h; a.w = w; a.Hs = Hs; a.Ws = Ws;
    a.attn_temp = attn_temp; a.sqrt_c = sqrtf((float)C); a.fuse_d = attn_fuse_d;
    ba.fwd_out = out; ba.wsum = wsum; ba.grad_out = grad_out; ba.grad_ref = grad_ref; ba.grad_src = grad_src;
    ba.windows = nullptr; ba.win_org = nullptr; ba.maxima = mx;
    ba.rec_offset = offset; ba.rec_cursor = cursor; ba.rec = rec; ba.tiles_x = tiles_x; ba.tiles_y = tiles_y;
    hipStream_t s = (hipStream_t)stream;
    // counts, cursors and the operand maxima start at zero: a kernel of our own, not hipMemsetAsync -- inside a captured
    // training step the memset node was seen to race with the kernels around it (garbage counts -> a memory fault on a
    // later replay); kernel nodes of one stream are strictly ordered
    hipLaunchKernelGGL(warp_bwd_zero_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, s, ints, 2 * n,
                       reinterpret_cast<int*>(mx), 4L);
    const long per_item = (long)h * w * D;
    const long per_b = (per_item + 255) / 256;
    {
        const long n_go = (long)B * D * h * w * G, n_ref = (long)h * w * C, n_src = (long)Hs * Ws * C;
        if (ref_batch_stride == n_ref && src_batch_stride == n_src && src_view_stride == n_src * B &&
            per_b * B < (1L << 30)) {
            // (contiguous operands: the maxima and the counting pass share a launch)
            const float* xs[3] = {grad_out, ref_feat, src_feat};
            const long ns[3] = {n_go, n_ref * B, n_src * B * NV};
            const AbsMaxArgs am = absmax_args(xs, ns, 3);
            hipLaunchKernelGGL(warp_bwd_prep_kernel, dim3((unsigned)(am.first[3] + per_b * B)), dim3(256), 0, s, a, count, tiles_x,
                               tiles_y, am, mx, am.first[3], (int)per_b);
        } else {
            launch_absmax(&grad_out, &n_go, 1, mx, s);
            for (int bb = 0; bb < B; ++bb) {
                const float* p = ref_feat + (long)bb * ref_batch_stride;
                launch_absmax(&p, &n_ref, 1, mx + 1, s);
            }
            for (int v = 0; v < NV; ++v)
                for (int bb = 0; bb < B; ++bb) {
                    const float* p = src_feat + (long)v * src_view_stride + (long)bb * src_batch_stride;
                    launch_absmax(&p, &n_src, 1, mx + 2, s);
                }
            hipLaunchKernelGGL(warp_bwd_count_kernel, dim3((unsigned)per_b, B), dim3(256), 0, s, a, count, tiles_x, tiles_y);
        }
    }
    hipLaunchKernelGGL(warp_bwd_scan_kernel, dim3(1), dim3(1024), 0, s, count, offset, (int)n);
    int rc = MVSTER_ERR_UNSUPPORTED;
#define MV_CASE(CC, GG, GR) \
    if (C == CC && G == GG && (group_cor != 0) == GR) rc = launch_bwd_rec<CC, GG, GR>(ba, s);
    MV_CASE(64, 8, true)
    MV_CASE(32, 8, true)
    MV_CASE(16, 4, true)
    MV_CASE(8, 4, true)
    MV_CASE(16, 8, true)
    MV_CASE(8, 8, true)
    MV_CASE(64, 4, true)
    MV_CASE(32, 4, true)
    MV_CASE(8, 8, false)
    MV_CASE(16, 16, false)
    MV_CASE(32, 32, false)
    MV_CASE(64, 64, false)
#undef MV_CASE
    if (rc) return rc;
    WarpBwdAccumArgs k2{rec, offset, mx, grad_src, src_view_stride, src_batch_stride, NV, C / 8, C, Hs, Ws, tiles_x, tiles_y,
                        G, D, C / G, group_cor ? 1 : 0, attn_fuse_d ? 1 : 0, attn_temp};
    hipLaunchKernelGGL(warp_bwd_accum_kernel, dim3(tiles_x * tiles_y, NV * (C / 8), B), dim3(256), 0, s, k2);
    return mv_check_launch();
}
