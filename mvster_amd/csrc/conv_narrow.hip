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


// ---- conv0a -> conv0b of the FPN in ONE launch (round 6) ---------------------------------------------------------------
// The two full-resolution layers of FPN4.conv0 (models/mvs4net_utils.py:427-428: 3 -> 8 -> 8, each conv + BatchNorm + ReLU,
// eval statistics folded into scale / shift) as two launches write and re-read the 8-channel intermediate: 26 + 52 MB and
// 52 + 52 MB at 5 x 512 x 640.  Here the intermediate lives in LDS: a workgroup computes the first layer on the
// (TY + 2) x (TX + 2) halo tile (zero where the tile leaves the image: the second layer's padding) and the second layer from it,
// 26 + 52 MB of HBM traffic in all.  Same MFMA packing as conv_narrow_kernel (N = 2 pixels x 8 channels) in both phases.
//   tile 14 x 64 output pixels; mid tile 16 x 66 = per row two 32-pixel units + ONE pixel pair, and the 16 rows' extra pairs
//   together are a 33rd unit (M index = row);  phase 1 = 33 units x 12 MFMAs, phase 2 = 28 units x 24: 1.19 MFMAs per output
//   pixel against 1.125 for the two launches (the halo costs 6 %).
//   LDS: ring of two input patches (18 x 68 pixels x 16 B, LDS-DMA by the four loading waves) + the mid tile (33 KB) = 74 KB,
//   two workgroups per CU.  Two barriers per tile: patch landed / mid free, mid complete / patch free.
// (probe build: bits of `dbg` knock out parts of the kernel for timing -- 1 / 2: the MFMAs of phase 1 / 2, 4: the output stores,
//  8: the input loads, 16: every tile, 32: the epilogues -- a template argument, so that a switch costs nothing where it is off;
//  scripts/conv_narrow_pair_bench.py)
#define PAIR_DBG DBG
struct NarrowPairArgs {
    const float* in;      // [NB, H, W, 4]
    const float* w1;      // [3][3][4][8]
    const float* scale1;  // [8]
    const float* shift1;
    const float* w2;      // [3][3][8][8]
    const float* scale2;
    const float* shift2;
    float* out;           // [NB, H, W, 8]
    int NB, H, W, relu1, relu2, dbg;
    unsigned in_bytes, out_bytes, ntiles;
    FastDiv tiles_x, tiles_y;
};

struct PairGeom {
    static constexpr int TY = 14, TX = 64;
    static constexpr int MH = TY + 2, MW = TX + 2;             // mid tile (first layer's outputs)
    static constexpr int PH = TY + 4, PW = TX + 4;             // input patch
    static constexpr int PSLOTS = PH * PW;                     // float4 slots (four channels per pixel)
    static constexpr int NIW = ((PSLOTS + 63) / 64 + 3) / 4;   // DMA wave-instructions per loading wave
    static constexpr int STAGE = NIW * 4 * 64;                 // slots of a ring stage (surplus slots included)
    static constexpr int R = 2;
    static constexpr int MID = MH * MW * 2;
    static constexpr size_t LDS = (size_t)(R * STAGE + MID + 8) * 16;   // + scale / shift of both layers
};

template <int DBG>
__global__ void __launch_bounds__(512, 4) conv_narrow_pair_kernel(NarrowPairArgs a) {
    using G = PairGeom;
    constexpr int TY = G::TY, PW = G::PW, MW = G::MW, NIW = G::NIW;
    static_assert(G::MH == 16, "the extra pixel pairs of the mid tile's rows form exactly one 16-row MFMA unit");
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const mid = lds + G::R * G::STAGE;

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const int lm = lane & 15, lq = lane >> 4;
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);
    if (PAIR_DBG & 16) tile = a.ntiles;

    auto decode = [&](unsigned t, int& nb, int& y0, int& x0) {
        unsigned txu, tyu;
        nb = (int)fdivmod(fdivmod(t, a.tiles_x, txu), a.tiles_y, tyu);
        y0 = (int)tyu * TY;
        x0 = (int)txu * G::TX;
    };

    if (loader) {
        const __amdgpu_buffer_rsrc_t in_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
        unsigned dbase[NIW];
        int dpos[NIW];                                          // column | row << 8; -1 = a surplus slot
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int s = (wave + 4 * n) * 64 + lane;
            const int prow = s / PW, pl = s - prow * PW;
            dpos[n] = s < G::PSLOTS ? (pl | (prow << 8)) : -1;
            dbase[n] = (unsigned)((prow * a.W + pl) * 16);
        }
        auto dma_tile = [&](unsigned t, int stage, bool live) {
            int nb, y0, x0;
            decode(t, nb, y0, x0);
            const unsigned porigin = (unsigned)((((nb * a.H + y0 - 2) * a.W) + x0 - 2) * 16);   // (32-bit wrap-around on borders)
            const unsigned wlim = live ? (unsigned)a.W : 0u;
            f32x4v* const dst0 = lds + stage * G::STAGE;
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const int ix = x0 + (dpos[n] & 255) - 2, iy = y0 + (dpos[n] >> 8) - 2;
                const bool ok = dpos[n] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < wlim && !(PAIR_DBG & 8);
                const unsigned off = ok ? dbase[n] + porigin : 0x80000000u;
                f32x4v* const dst = dst0 + (wave + 4 * n) * 64;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
            }
        };
        dma_tile(tile < a.ntiles ? tile : 0u, 0, tile < a.ntiles);
        {
            const unsigned t = tile + nwg;
            dma_tile(t < a.ntiles ? t : 0u, 1, t < a.ntiles);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");          // the first patch has landed
        int st = 0;
        for (; tile < a.ntiles; tile += nwg) {
            __builtin_amdgcn_s_barrier();                       // (1) patch of this tile visible
            __builtin_amdgcn_s_barrier();                       // (2) phase 1 has read it: its stage is free
            const unsigned t = tile + 2u * nwg;
            dma_tile(t < a.ntiles ? t : 0u, st, t < a.ntiles);
            st ^= 1;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");      // the next tile's patch has landed
        }
        return;
    }

    // ---- compute waves -------------------------------------------------------------------------------------------------
    // Work split: wave w = (row group w >> 1, half w & 1).  Phase 1: mid rows 8 g .. 8 g + 7 of its 32-pixel half, two rows per
    // pass; phase 2: output rows 7 g .. 7 g + 6 of its half, two rows per pass and a last single row.  The two rows of a pass
    // share operand rows (4 instead of 6 row reads per pass), consecutive passes share two more, and the rows a pass adds are
    // read while the pass before it computes.  A pass's epilogue (scale / shift / ReLU, mid or global store) is issued in the
    // MFMA shadow of the NEXT pass -- measured with the probe build's knock-outs, the one-pass-at-a-time form spent 23 us in
    // MFMAs at their full rate and 20 us beside them in epilogues, operand latency and barriers, nothing overlapping.
    const int delta = lm >> 3, co = lm & 7;
    const int half = wave & 1, grp = wave >> 1;
    // layer 1 (four input channels): K step s = tap row ky, K slot = (channel j, tap column kx' = lq)
    f32x4v wf1[3];
    const int p1 = 32 * half + 2 * lm + lq;                     // operand slot in a patch row
    const int px1 = lm * PW + 64 + lq;                          // the extra unit: M index = mid row, pixel pair (64, 65)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int kx = lq - delta;
        const bool nz = kx >= 0 && kx <= 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = a.w1[nz ? ((s * 3 + kx) * 4 + j) * 8 + co : 0];
            wf1[s][j] = nz ? v : 0.0f;
        }
    }
    // layer 2 (eight input channels): K step s = (tap row s >> 1, tap-column pair e = s & 1); K slot = (tap column
    // kx' = 2 e + (lq >> 1), channels (lq & 1) * 4 + j)
    f32x4v wf2[6];
    int p2[2];                                                  // operand slots in a mid row, e = 0, 1
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int pxl = 2 * lm + 2 * e + (lq >> 1);
        p2[e] = 64 * half + (pxl ^ ((pxl >> 3) & 1)) * 2 + (lq & 1);
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int t = 2 * s + (lq >> 1);
        const int ky = t >> 2, kxp = t & 3, c0 = (lq & 1) * 4;
        const int kx = kxp - delta;
        const bool nz = kx >= 0 && kx <= 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = a.w2[nz ? ((ky * 3 + kx) * 8 + c0 + j) * 8 + co : 0];
            wf2[s][j] = nz ? v : 0.0f;
        }
    }
    // scale / shift of both layers wait in LDS (eight float4 slots behind the mid tile), read once per pass
    f32x4v* const consts = mid + G::MID;
    if (wave == 0 && lane < 32) {
        const float* const src = lane < 8 ? a.scale1 : lane < 16 ? a.shift1 : lane < 24 ? a.scale2 : a.shift2;
        reinterpret_cast<float*>(consts)[lane] = src[lane & 7];
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // (visible to the other waves after the first barrier)
    }
    const int hi = lq & 1;
    const float floor1 = a.relu1 ? 0.0f : -__builtin_inff(), floor2 = a.relu2 ? 0.0f : -__builtin_inff();
    // accumulators are D^T: lane (lm, lq) holds channels (lq & 1) * 4 .. + 3 of pixel 2 lm + (lq >> 1) of its unit
    const int col = 32 * half + 2 * lm + (lq >> 1);             // column in the tile (phase 2) / in the mid tile (phase 1)
    const int pc = 2 * lm + (lq >> 1);
    const int mcol = 64 * half + (pc ^ ((pc >> 3) & 1)) * 2 + hi;   // its slot in a mid row (pair swap of the 8-channel layout)
    const unsigned ocol = (unsigned)((col * 8 + hi * 4) * 4);   // byte offset of its 16 bytes in an output row
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)a.out_bytes, 0x00020000);
    int st = 0;
    unsigned turn = 0;                                          // the extra unit goes round the waves
    for (; tile < a.ntiles; tile += nwg, ++turn) {
        int nb, y0, x0;
        decode(tile, nb, y0, x0);
        const f32x4v* const stage = lds + st * G::STAGE;
        st ^= 1;
        const bool colok1 = (unsigned)(x0 - 1 + col) < (unsigned)a.W;       // this lane's mid pixel lies inside the image
        const bool colok2 = x0 + col < a.W;
        __builtin_amdgcn_s_barrier();                           // (1) patch landed; everyone is done with the old mid tile
        // ---- phase 1: mid = ReLU(BN(conv(in))) on the halo tile, zero outside the image --------------------------------
        {
            const int R0 = grp * 8;
            const f32x4v* const xs = stage + R0 * PW + p1;
            f32x4v* const ms = mid + R0 * MW * 2 + mcol;
            f32x4v X[4], N[2], pend[2], sc, sh;
#pragma unroll
            for (int i = 0; i < 4; ++i) X[i] = xs[i * PW];
            // one epilogue step per MFMA slot: q = 0 reads scale / shift, 2 .. 5 finish the pass's two rows (mid rows R0 + row ..)
            auto epi1 = [&](int q, int row) {
                if (PAIR_DBG & 32) return;
                if (q == 0) { sc = consts[hi]; sh = consts[2 + hi]; }
                if (q == 2 || q == 4) {
                    f32x4v& v = pend[(q - 2) >> 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), floor1);
                }
                if (q == 3 || q == 5) {
                    const int m = (q - 3) >> 1;
                    const bool in_img = colok1 && (unsigned)(y0 - 1 + R0 + row + m) < (unsigned)a.H;
                    ms[(row + m) * MW * 2] = in_img ? pend[m] : (f32x4v){0.f, 0.f, 0.f, 0.f};
                }
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < 3) {
                    N[0] = xs[(2 * k + 4) * PW];
                    N[1] = xs[(2 * k + 5) * PW];
                }
                f32x4v acc[2];
                acc[0] = acc[1] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!(PAIR_DBG & 1)) {
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[s][j], X[s][j], acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[s][j], X[s + 1][j], acc[1], 0, 0, 0);
                        }
                        if (k > 0) epi1(s * 4 + j, 2 * (k - 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                pend[0] = acc[0];
                pend[1] = acc[1];
                X[0] = X[2];
                X[1] = X[3];
                X[2] = N[0];
                X[3] = N[1];
            }
            if (wave == (int)(turn & 3)) {
                // the 33rd unit (pixel pair (64, 65) of all 16 mid rows) with the last pass's epilogue in its shadow
                f32x4v E[3], acc = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 3; ++s) E[s] = stage[s * PW + px1];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!(PAIR_DBG & 1)) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[s][j], E[s][j], acc, 0, 0, 0);
                        epi1(s * 4 + j, 6);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                // lane (lm, lq): channels (lq & 1) * 4 .. of pixel 64 + (lq >> 1) of mid row lm
                if (!(PAIR_DBG & 32)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = fmaxf(fmaf(acc[j], sc[j], sh[j]), floor1);
                    const bool in_img = (unsigned)(y0 - 1 + lm) < (unsigned)a.H && (unsigned)(x0 + 63 + (lq >> 1)) < (unsigned)a.W;
                    mid[lm * MW * 2 + (64 + (lq >> 1)) * 2 + hi] = in_img ? acc : (f32x4v){0.f, 0.f, 0.f, 0.f};
                }
            } else {
#pragma unroll
                for (int q = 0; q < 6; ++q) epi1(q, 6);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): mid written, patch read
        __builtin_amdgcn_s_barrier();                           // (2)
        // ---- phase 2: out = ReLU(BN(conv(mid))) ------------------------------------------------------------------------
        {
            const int Q0 = grp * 7;
            const f32x4v* const ys = mid + Q0 * MW * 2;
            const unsigned oorigin = (unsigned)((((nb * a.H + y0 + Q0) * a.W) + x0) * 32) + ocol;
            f32x4v Y[4][2], N[2][2], pend[2], sc, sh;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e) Y[i][e] = ys[i * MW * 2 + p2[e]];
            auto epi2 = [&](int q, int row, int nrows) {
                if (PAIR_DBG & 32) return;
                if (q == 0) { sc = consts[4 + hi]; sh = consts[6 + hi]; }
                if (q == 2 || (q == 4 && nrows == 2)) {
                    f32x4v& v = pend[(q - 2) >> 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(v[j], sc[j], sh[j]), floor2);
                }
                if (q == 3 || (q == 5 && nrows == 2)) {
                    const int m = (q - 3) >> 1;
                    const bool ok = colok2 && y0 + Q0 + row + m < a.H && !(PAIR_DBG & 4);
                    const unsigned off = ok ? oorigin + (unsigned)((row + m) * a.W * 32) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, pend[m]), out_rsrc, off, 0, MV_STORE_AUX);
                }
            };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // rows 2 k, 2 k + 1 (k = 3: the single row 6): mid rows Q0 + 2 k .. + 3 are in Y; read what the next pass adds
                if (k < 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 2; ++e) N[i][e] = ys[(2 * k + 4 + i) * MW * 2 + p2[e]];
                } else if (k == 2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) N[0][e] = ys[8 * MW * 2 + p2[e]];
                }
                f32x4v acc[2];
                acc[0] = acc[1] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 6; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!(PAIR_DBG & 2)) {
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf2[s][j], Y[s >> 1][s & 1][j], acc[0], 0, 0, 0);
                            if (k < 3)
                                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf2[s][j], Y[(s >> 1) + 1][s & 1][j], acc[1], 0, 0, 0);
                        }
                        if (k > 0) epi2(s * 4 + j, 2 * (k - 1), 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                pend[0] = acc[0];
                pend[1] = acc[1];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    Y[0][e] = Y[2][e];
                    Y[1][e] = Y[3][e];
                    Y[2][e] = N[0][e];
                    Y[3][e] = N[1][e];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) epi2(q, 6, 1);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): done reading the mid tile
    }
}

int launch_narrow_pair(NarrowPairArgs& a, int wpc, hipStream_t s) {
    using G = PairGeom;
    auto kern = conv_narrow_pair_kernel<0>;
    int slot = 0;
#ifdef MVSTER_PROBES
    switch (a.dbg) {
#define PAIR_CASE(n, m) case m: kern = conv_narrow_pair_kernel<m>; slot = n; break;
        PAIR_CASE(1, 1) PAIR_CASE(2, 2) PAIR_CASE(3, 3) PAIR_CASE(4, 4) PAIR_CASE(5, 8) PAIR_CASE(6, 12) PAIR_CASE(7, 15)
        PAIR_CASE(8, 16) PAIR_CASE(9, 32) PAIR_CASE(10, 35) PAIR_CASE(11, 47) PAIR_CASE(12, 44)
#undef PAIR_CASE
        case 0: break;
        default: return MVSTER_ERR_UNSUPPORTED;
    }
#endif
    static unsigned long attr_done_all[13] = {0};
    unsigned long& attr_done = attr_done_all[slot];
    if (G::LDS > 64 * 1024 && !mvconv::allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = mvconv::num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    const unsigned tx = (unsigned)((a.W + G::TX - 1) / G::TX), ty = (unsigned)((a.H + G::TY - 1) / G::TY);
    const long ntiles = (long)tx * ty * a.NB;
    if (ntiles >= (1L << 30)) return MVSTER_ERR_SHAPE;
    a.ntiles = (unsigned)ntiles;
    a.tiles_x = mv_fastdiv(tx);
    a.tiles_y = mv_fastdiv(ty);
    const int by_lds = (int)((160 * 1024) / G::LDS);
    int per_cu = wpc > 0 ? wpc : 2;
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    const long gmax = (long)ncu * per_cu;
    const long rounds = (ntiles + gmax - 1) / gmax;            // equal shares
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv_narrow_pair_kernel");
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

// FPN4.conv0 in one launch (models/mvs4net_utils.py:427-428): in [NB,H,W,4] (RGB0) -> conv3x3 (w1 [3][3][4][8]) * scale1 +
// shift1, ReLU if relu1 -> conv3x3 (w2 [3][3][8][8]) * scale2 + shift2, ReLU if relu2 -> out [NB,H,W,8].  The 8-channel
// intermediate never reaches HBM.  wpc: workgroups per CU (0 = default, 2).
extern "C" int mvster_conv_narrow_pair(const float* in, const float* w1, const float* scale1, const float* shift1,
                                       const float* w2, const float* scale2, const float* shift2, float* out, int NB, int H,
                                       int W, int relu1, int relu2, int wpc, void* stream) {
    if (!in || !w1 || !scale1 || !shift1 || !w2 || !scale2 || !shift2 || !out) return MVSTER_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0) return MVSTER_ERR_SHAPE;
    const long in_bytes = (long)NB * H * W * 16, out_bytes = (long)NB * H * W * 32;
    if (out_bytes >= (1L << 31)) return MVSTER_ERR_SHAPE;
    NarrowPairArgs a;
    a.in = in; a.w1 = w1; a.scale1 = scale1; a.shift1 = shift1; a.w2 = w2; a.scale2 = scale2; a.shift2 = shift2; a.out = out;
    a.NB = NB; a.H = H; a.W = W; a.relu1 = relu1; a.relu2 = relu2;
    a.in_bytes = (unsigned)in_bytes; a.out_bytes = (unsigned)out_bytes;
    a.dbg = wpc >> 4;                                           // (read by the probe build only)
    return launch_narrow_pair(a, wpc & 15, (hipStream_t)stream);
}
