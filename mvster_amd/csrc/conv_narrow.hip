// 3x3 stride-1 convolution of the narrow full-resolution layers (Cin in {4, 8} -> 8 channels) on the fp32 matrix cores,
// N operand packed by a pixel shift, on a persistent LDS-DMA ring (round 4).
//
// Why: as a plain implicit GEMM these layers (FPN conv0.0 / conv0.1, the composed FPN tail, conv0 of every reg2d) fill only
// 8 of the 16 columns of a v_mfma_f32_16x16x4_f32 tile, which costs exactly what the packed-FMA VALU kernel
// (conv_small_kernel) costs: 36 SIMD cycles per pixel for 8 -> 8 -- 24-29 us for the five 512x640 maps of a forward against
// 17 us of HBM time.  Here the empty half of N holds the NEXT pixel's outputs: row (delta, co) of the weight operand is
// W[ky][kx' - delta][ci][co] over a K axis that spans FOUR tap columns kx' = 0..3 (zero where kx' - delta leaves 0..2), so
// one 16 x 16 tile = 16 pixel PAIRS x (2 pixels x 8 channels) and K = 3 * 4 * Cin: 24 (12) MFMAs per 32 pixels instead of
// 36 (18) -- 24 SIMD cycles per pixel, under the HBM time (algebra: scripts/probes/narrow_conv_shift_packing.py).
//
// Frame: workgroup = 4 compute waves + 4 loading waves, persistent over its share of the TY x 32-pixel tiles (TY = 4 MT
// rows: wave w owns rows w*MT ..).  The loading waves keep a ring of R stages filled by LDS-DMA (buffer_load ... lds, zero
// padding = the descriptor's range check), R - 1 tiles ahead of the compute waves: a CU needs ~40 KB in flight to stream at
// its share of the HBM rate, one tile (11 KB) is not enough.  A stage = the (TY + 2) x 34-pixel input patch and, with SKIP, the
// tile of the tensor added in the epilogue (so the compute waves never wait on global memory).  One barrier per tile.
// LDS layout of the patch: [row][pixel][quad]; for Cin = 8 the two pixels of every pair whose index has bit 2 set are
// swapped (a source-address permutation of the lane-linear DMA), which makes the 16-lane groups of the ds_read_b128
// operand reads conflict-free; Cin = 4 needs no permutation.
// K order = (ky, kx', ci) with ci in the fragment permutation; products with the structural zeros are exact no-ops for
// finite inputs (a non-finite input pixel reaches one more output column than in the reference: 0 * inf).
// Reference layers: models/mvs4net_utils.py:427-428 (FPN conv0), :875 (reg2d conv0), :459 (out4, composed, section 4.3).
#include "conv_args.hpp"

namespace {

using mvconv::f32x4v;
using mvconv::lds_void;
using mvconv::u32x4v;

struct NarrowArgs {
    const float* in;     // [NB, H, W, CIN]
    const float* w;      // [3][3][CIN][8]
    const float* scale;  // [8]
    const float* shift;  // [8]
    const float* skip;   // [NB, H, W, 8] or null
    float* out;          // [NB, H, W, 8]
    int NB, H, W, relu;
    unsigned in_bytes, out_bytes, ntiles;
    FastDiv tiles_x, tiles_y;
};

template <int CIN, int MT, int R, bool SKIP>
struct NarrowGeom {
    static constexpr int Q = CIN / 4;                          // float4 slots per pixel
    static constexpr int TY = 4 * MT, PH = TY + 2, PW = 34;
    static constexpr int PSLOTS = PH * PW * Q;
    static constexpr int NBLK = (PSLOTS + 63) / 64;            // DMA wave-instructions of the patch
    static constexpr int SBLK = SKIP ? TY : 0;                 // ... of the skip tile (TY rows x 32 pixels x 2 quads)
    static constexpr int NI = NBLK + SBLK;
    static constexpr int NIW = (NI + 3) / 4;                   // per loading wave
    static constexpr int STAGE = NI * 64;                      // float4 slots of a ring stage
    static constexpr size_t LDS = (size_t)(R * STAGE + 64) * 16;
};

// CO4: the layer has FOUR output channels (the input gradient of reg2d's first layer in training: 8 -> 4, the cost volume's
// groups): w is padded to eight with zeros, the lanes of channels 4..7 store nothing and the output pitch is four.
template <int CIN, int MT, int R, bool SKIP, bool CO4 = false>
__global__ void __launch_bounds__(512) conv_narrow_kernel(NarrowArgs a) {
    static_assert(!(CO4 && SKIP), "the four-channel form has no skip path");
    using G = NarrowGeom<CIN, MT, R, SKIP>;
    constexpr int Q = G::Q, TY = G::TY, PW = G::PW, NBLK = G::NBLK, NI = G::NI, NIW = G::NIW;
    constexpr int NKS = 3 * Q;                                 // K steps of 16: (tap pair x 8 channels) or (tap row x 4 channels)
    static_assert((R - 2) * NIW <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const scratch = lds + R * G::STAGE;                // target of the surplus DMA slots

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const int lm = lane & 15, lq = lane >> 4;
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);                // this workgroup's tiles: tile, tile + nwg, ...

    auto decode = [&](unsigned t, int& nb, int& y0, int& x0) {
        unsigned txu, tyu;
        nb = (int)fdivmod(fdivmod(t, a.tiles_x, txu), a.tiles_y, tyu);
        y0 = (int)tyu * TY;
        x0 = (int)txu * 32;
    };

    if (loader) {
        const __amdgpu_buffer_rsrc_t in_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)a.out_bytes : 0, 0x00020000);
        // instruction i = wave + 4n: i < NBLK -> 64 slots of the patch, NBLK <= i < NI -> one row of the skip tile.
        // dbase = byte offset of the lane's 16 bytes relative to the patch (skip tile) origin, 0x80000000 = no pixel;
        // dpos = column | row << 8 for the border tests.
        unsigned dbase[NIW];
        int dpos[NIW];
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int i = wave + 4 * n;
            if (i < NBLK) {
                const int s = i * 64 + lane;
                const int quad = s % Q, pp = s / Q;
                const int prow = pp / PW, pl = pp - prow * PW;
                const int px = CIN == 8 ? pl ^ ((pl >> 3) & 1) : pl;          // pixel held by LDS position pl
                dpos[n] = px | (prow << 8);
                dbase[n] = s < G::PSLOTS ? (unsigned)(((prow * a.W + px) * CIN + quad * 4) * 4) : 0x80000000u;
            } else {
                const int s = (i - NBLK) * 64 + lane;
                const int row = s >> 6, x = (s >> 1) & 31, quad = s & 1;
                dpos[n] = x | (row << 8);
                dbase[n] = i < NI ? (unsigned)(((row * a.W + x) * 8 + quad * 4) * 4) : 0x80000000u;
            }
        }
        auto dma_tile = [&](unsigned t, int stage, bool live) {
            int nb, y0, x0;
            decode(t, nb, y0, x0);
            // (may be "negative" on border tiles: 32-bit wrap-around arithmetic)
            const unsigned porigin = (unsigned)((((nb * a.H + y0 - 1) * a.W) + x0 - 1) * (CIN * 4));
            const unsigned sorigin = (unsigned)((((nb * a.H + y0) * a.W) + x0) * 32);
            const unsigned wlim = live ? (unsigned)a.W : 0u;
            f32x4v* const dst0 = lds + stage * G::STAGE;
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const int i = wave + 4 * n;
                const bool sk = SKIP && i >= NBLK;
                const int ix = x0 + (dpos[n] & 255) - (sk ? 0 : 1), iy = y0 + (dpos[n] >> 8) - (sk ? 0 : 1);
                const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < wlim;
                // (named operand: hipcc 7.2 drops the kernel's host stub when this builtin gets an expression as its offset)
                const unsigned off = ok ? dbase[n] + (sk ? sorigin : porigin) : 0x80000000u;
                f32x4v* const dst = i < NI ? dst0 + i * 64 : scratch;
                if (sk) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(skip_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                }
            }
        };
        // fill R - 1 stages, then stay R - 1 tiles ahead; vmcnt counts this wave's requests in order, so "(R - 2) tiles'
        // worth outstanding" = the oldest tile in flight has landed
#pragma unroll
        for (int k = 0; k < R - 1; ++k) {
            const unsigned t = tile + (unsigned)k * nwg;
            dma_tile(t < a.ntiles ? t : 0u, k, t < a.ntiles);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * NIW) : "memory");
        __builtin_amdgcn_s_barrier();
        int st = R - 1;
        for (; tile < a.ntiles; tile += nwg) {
            const unsigned t = tile + (unsigned)(R - 1) * nwg;
            dma_tile(t < a.ntiles ? t : 0u, st, t < a.ntiles);
            st = st + 1 == R ? 0 : st + 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * NIW) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---- compute waves -------------------------------------------------------------------------------------------------
    // weight fragments: row lm = (delta, co) of the packed operand, K slot (s, j, lq)
    f32x4v wfrag[NKS];
    int toff[NKS];
    {
        const int delta = lm >> 3, co = lm & 7;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            int ky, kxp, c0;
            if (CIN == 8) {
                const int t = 2 * s + (lq >> 1);
                ky = t >> 2; kxp = t & 3; c0 = (lq & 1) * 4;
                const int pxl = 2 * lm + kxp;
                toff[s] = (ky * PW + (pxl ^ ((pxl >> 3) & 1))) * 2 + (lq & 1);
            } else {
                ky = s; kxp = lq; c0 = 0;
                toff[s] = ky * PW + 2 * lm + kxp;
            }
            const int kx = kxp - delta;
            const bool nz = kx >= 0 && kx <= 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = a.w[nz ? ((ky * 3 + kx) * CIN + c0 + j) * 8 + co : 0];
                wfrag[s][j] = nz ? v : 0.0f;
            }
        }
    }
    const f32x4v scv = *reinterpret_cast<const f32x4v*>(a.scale + (lq & 1) * 4);
    const f32x4v shv = *reinterpret_cast<const f32x4v*>(a.shift + (lq & 1) * 4);
    // the accumulator is D^T (weights in the A slot): lane (lm, lq) ends up with channels (lq & 1) * 4 .. + 3 of pixel
    // 2 lm + (lq >> 1) of its row
    const int col = 2 * lm + (lq >> 1);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)a.out_bytes, 0x00020000);
    __builtin_amdgcn_s_barrier();                               // the first stage has landed
    int st = 0;
    for (; tile < a.ntiles; tile += nwg) {
        int nb, y0, x0;
        decode(tile, nb, y0, x0);
        const f32x4v* const stage = lds + st * G::STAGE;
        st = st + 1 == R ? 0 : st + 1;
        f32x4v acc[MT], skv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            skv[mt] = SKIP ? stage[NBLK * 64 + (wave * MT + mt) * 64 + 4 * lm + lq] : (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
        // two rows per pass: all 2 * NKS operand reads of a pass are in flight before its first MFMA (left to itself hipcc
        // sinks every ds_read next to its use and waits out the LDS latency 2 * NKS times per tile)
#pragma unroll
        for (int h = 0; h < MT; h += 2) {
            f32x4v A[2][NKS];
#pragma unroll
            for (int s = 0; s < NKS; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m) A[m][s] = stage[(wave * MT + h + m) * PW * Q + toff[s]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NKS; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // (pinned order: the two rows' accumulators alternate, so that no MFMA waits for its predecessor's result --
                    //  left alone hipcc issues runs of up to eight dependent ones, 40 instead of 32 cycles each)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[h + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag[s][j], A[m][s][j], acc[h + m], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        const unsigned oorigin = (unsigned)((((nb * a.H + y0) * a.W) + x0) * (CO4 ? 16 : 32));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = wave * MT + mt;
            f32x4v v = acc[mt];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = fmaf(v[j], scv[j], shv[j]);
                if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                if (SKIP) v[j] += skv[mt][j];
            }
            const bool ok = y0 + row < a.H && x0 + col < a.W && !(CO4 && (lq & 1));
            const unsigned off = ok ? oorigin + (unsigned)(((row * a.W + col) * (CO4 ? 4 : 8) + (lq & 1) * 4) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, off, 0, MV_STORE_AUX);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): done reading this stage
        __builtin_amdgcn_s_barrier();
    }
}

template <int CIN, int MT, int R, bool SKIP, bool CO4 = false>
int launch_narrow(NarrowArgs& a, int wpc, hipStream_t s) {
    using G = NarrowGeom<CIN, MT, R, SKIP>;
    auto kern = conv_narrow_kernel<CIN, MT, R, SKIP, CO4>;
    static unsigned long attr_done = 0;
    if (G::LDS > 64 * 1024 && !mvconv::allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = mvconv::num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    const unsigned tx = (unsigned)((a.W + 31) / 32), ty = (unsigned)((a.H + G::TY - 1) / G::TY);
    const long ntiles = (long)tx * ty * a.NB;
    if (ntiles >= (1L << 30)) return MVSTER_ERR_SHAPE;
    a.ntiles = (unsigned)ntiles;
    a.tiles_x = mv_fastdiv(tx);
    a.tiles_y = mv_fastdiv(ty);
    const int by_lds = (int)((160 * 1024) / G::LDS);
    int per_cu = wpc > 0 ? wpc : 2;
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu > 2) per_cu = 2;                                // (16 waves per workgroup pair = 4 per SIMD)
    if (per_cu < 1) per_cu = 1;
    const long gmax = (long)ncu * per_cu;
    const long rounds = (ntiles + gmax - 1) / gmax;            // equal shares
    const long gx = (ntiles + rounds - 1) / rounds;
    if (CO4) {
        MV_NOTE_KERNEL("conv_narrow_kernel<%d, %d, %d, false, true>", CIN, MT, R);
    } else {
        MV_NOTE_KERNEL("conv_narrow_kernel<%d, %d, %d, %s>", CIN, MT, R, SKIP ? "true" : "false");
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), G::LDS, s, a);
    return mv_check_launch();
}

}  // namespace

// in [NB,H,W,cin] channels-last (cin in {4, 8}), w [3][3][cin][8], scale / shift [8], skip [NB,H,W,8] or null ->
// out [NB,H,W,8] = (conv3x3(in) * scale + shift, ReLU if relu) + skip.  mt: tile rows / 4 (2 or 4; 0 = by size);
// wpc: workgroups per CU (0 = default).  Same contract as mvster_conv_small (the VALU form of the same layers).
extern "C" int mvster_conv_narrow(const float* in, const float* w, const float* scale, const float* shift,
                                  const float* skip, float* out, int NB, int H, int W, int cin, int relu, int mt,
                                  int wpc, void* stream) {
    if (!in || !w || !scale || !shift || !out) return MVSTER_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0) return MVSTER_ERR_SHAPE;
    if (cin != 4 && cin != 8) return MVSTER_ERR_UNSUPPORTED;
    const long in_bytes = (long)NB * H * W * cin * 4, out_bytes = (long)NB * H * W * 32;
    if (in_bytes >= (1L << 31) || out_bytes >= (1L << 31)) return MVSTER_ERR_SHAPE;
    NarrowArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.skip = skip; a.out = out;
    a.NB = NB; a.H = H; a.W = W; a.relu = relu;
    a.in_bytes = (unsigned)in_bytes; a.out_bytes = (unsigned)out_bytes;
    if (mt == 0) mt = (long)NB * H * W >= (1L << 20) ? 4 : 2;
    hipStream_t s = (hipStream_t)stream;
    if (cin == 8) {
        if (mt == 2) return skip ? launch_narrow<8, 2, 4, true>(a, wpc, s) : launch_narrow<8, 2, 4, false>(a, wpc, s);
        if (mt == 4) return skip ? launch_narrow<8, 4, 3, true>(a, wpc, s) : launch_narrow<8, 4, 3, false>(a, wpc, s);
    } else {
        if (mt == 2) return skip ? launch_narrow<4, 2, 4, true>(a, wpc, s) : launch_narrow<4, 2, 4, false>(a, wpc, s);
        if (mt == 4) return skip ? launch_narrow<4, 4, 4, true>(a, wpc, s) : launch_narrow<4, 4, 4, false>(a, wpc, s);
    }
    return MVSTER_ERR_UNSUPPORTED;
}

// The 8 -> 4 form: in [NB,H,W,8], w [3][3][8][8] with output columns 4..7 zero, scale / shift [8] (entries 4..7 unused) ->
// out [NB,H,W,4].  No skip.  The input gradient of reg2d's conv0 (models/mvs4net_utils.py:875) in training: the gradient of
// the cost volume's four groups, which ran on the direct kernel (129 us at [2, 4, 512, 640] for 31 us of HBM time).
extern "C" int mvster_conv_narrow4(const float* in, const float* w, const float* scale, const float* shift, float* out, int NB,
                                   int H, int W, int relu, int mt, int wpc, void* stream) {
    if (!in || !w || !scale || !shift || !out) return MVSTER_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0) return MVSTER_ERR_SHAPE;
    const long in_bytes = (long)NB * H * W * 32, out_bytes = (long)NB * H * W * 16;
    if (in_bytes >= (1L << 31)) return MVSTER_ERR_SHAPE;
    NarrowArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.skip = nullptr; a.out = out;
    a.NB = NB; a.H = H; a.W = W; a.relu = relu;
    a.in_bytes = (unsigned)in_bytes; a.out_bytes = (unsigned)out_bytes;
    if (mt == 0) mt = (long)NB * H * W >= (1L << 20) ? 4 : 2;
    hipStream_t s = (hipStream_t)stream;
    if (mt == 2) return launch_narrow<8, 2, 4, false, true>(a, wpc, s);
    if (mt == 4) return launch_narrow<8, 4, 3, false, true>(a, wpc, s);
    return MVSTER_ERR_UNSUPPORTED;
}
