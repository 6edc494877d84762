// Persistent, LDS-DMA double-buffered implicit-GEMM convolution (variant 5 of mvster_conv_mfma).
//
// Why: s_memtime stamps of conv_lds_kernel (profiles/r03_a_conv_lds_timeline.txt) show a wavefront of a one-chunk
// 16 -> 16 layer alive for ~19 500 cycles of which the 72 MFMAs are 29 % (5 700 cycles against 2 304 of pipe time):
// 32 % goes into the prologue (kernel arguments, tile decode, staging addresses, weight / BatchNorm fetches -- paid per
// 128-pixel tile), 19 % into waiting for the staged patch and the barrier, 20 % into the epilogue.  This kernel keeps a
// workgroup alive over many tiles:
//   * weights, scale/shift and all tile-independent index arithmetic are set up ONCE per workgroup; the weights stay
//     in LDS (or, for the 16 -> 16 3x3 layers, in 36 registers) for its lifetime;
//   * the input patch of tile t+1 is fetched by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass,
//     zero padding = the descriptor's range check) into the second LDS buffer while tile t's MFMAs run;
//   * one barrier per tile; the epilogue is a specialised float4 path (scale/shift, ReLU, optional same-shape skip).
// Same packed weights and fused epilogue semantics as the other variants; K order = tap-major, channel-minor (the packed
// order), i.e. bit-identical to the direct kernel.
//
// Workgroup = 4 waves = TY x 32 output pixels of one (b, z) slice, TY = 2*MT; wave w owns M tiles w*MT .. w*MT+MT-1
// (tile t = row t >> 1, columns (t & 1)*16 ..).  LDS patch layout per 16-channel chunk: two planes (plane = quad >> 1) of
// [pixel][2 quads] float4 -- the conflict-free layout of conv_lds_kernel; for stride-2 layers the pixels of a row are
// split into even / odd columns so that a tap again reads 16 consecutive pixels.  LDS-DMA writes are lane-linear
// (64 consecutive float4 per wave instruction), so the layout is produced by the SOURCE address decode.
//
// Reference layers: Conv2d / ConvBnReLU3D of models/mvs4net_utils.py:116-123, :224-251 as used by FPN4 (:419-502) and
// reg2d / reg3d (:870-965).
#include "conv_args.hpp"

namespace mvconv {
namespace {

typedef __attribute__((address_space(3))) void lds_void;

// Probe build only (make timeline; scripts/conv_pers_timeline.py): per wavefront and tile, s_memtime at the phase
// boundaries.  Record = 8 x u64 per (workgroup, wave, tile slot < 16): t0 loop top, t1 DMA of the next tile issued,
// t2 MFMAs issued, t3 past the barrier (next patch landed), t4 epilogue stores issued, [5] HW_ID, [6] XCC_ID, [7] tile.
#ifdef MVSTER_TIMELINE
__device__ unsigned long long* g_ptl = nullptr;
#define MV_PTL(k)                                                                                                       \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (g_ptl && lane == 0 && it < 16)                                                                              \
            g_ptl[(((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16 + it) * 8 + (k)] =                    \
                (k) < 5 ? __builtin_amdgcn_s_memtime() : (k) == 5 ? __builtin_amdgcn_s_getreg(63492)                    \
                                                         : (k) == 6 ? __builtin_amdgcn_s_getreg(63508) : tile;          \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define MV_PTL(k)
#endif

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct PersArgs {
    unsigned tiles_x, tiles_y, ntiles;       // tiles per (b, z) slice and in total
    unsigned out_bytes;                      // size of `out` (and of a same-shape skip): < 2^31
    unsigned mul[3], shr[3];                 // multiply-shift division by tiles_x, tiles_y, Do
};

struct TilePos { int b, zo, ty0, tx0; };

template <int MT, int KW, int SW, int KD>
struct PersGeom {
    static constexpr int TY = 2 * MT;
    static constexpr int KH = KW;
    static constexpr int PW = 31 * SW + KW;                  // patch width (input pixels)
    static constexpr int PH = (TY - 1) * SW + KH;            // patch height of one depth slice
    static constexpr int ROWS = KD * PH;
    static constexpr int PWH = (PW + 1) / 2;                 // stride 2: columns per parity
    static constexpr int ROWSLOTS = SW == 1 ? PW * 2 : PWH * 4;           // float4 slots of one patch row in one plane
    static constexpr int USED = ROWS * ROWSLOTS;                          // slots of one plane that hold pixels
    static constexpr int NBLK = (USED + 63) / 64;                         // DMA wave-instructions per plane
    static constexpr int PLANE = ((NBLK * 64 + 7) & ~7) + 4;              // plane pitch (float4), = 4 mod 8
};

// MT, NT: register tile (M tiles x N tiles of 16) per wave; KW = KH in {3, 5}; SW = SH in {1, 2}; NCH = cin / 16;
// KD in {1, 3}; WREG: the layer's weights live in registers (KD*KW*KW*NCH*NT float4 per lane), else in LDS;
// PF = prefetch distance of the tap pipeline (taps).
template <int MT, int NT, int KW, int SW, int NCH, int KD, bool WREG, int PF>
__global__ void __launch_bounds__(256) conv_pers_kernel(ConvArgs a, PersArgs p) {
    using G = PersGeom<MT, KW, SW, KD>;
    constexpr int TY = G::TY, KH = G::KH, PW = G::PW, PH = G::PH, PWH = G::PWH, PLANE = G::PLANE, NBLK = G::NBLK;
    constexpr int NTAP = KD * KH * KW, TAPS2D = KH * KW;
    constexpr int BUF = NCH * 2 * PLANE;                    // float4 per patch buffer
    constexpr int NI = NCH * 2 * NBLK;                      // DMA wave-instructions per tile
    constexpr int NIW = (NI + 3) / 4;
    constexpr int CIN = NCH * 16;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const wl = lds + 2 * BUF;                       // [tap][chunk][nt][lane]  (unused with WREG)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.y * NT;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);

    // ---- tile-independent per-lane part of the DMA addresses ------------------------------------------------------
    // instruction i = wave + 4n -> (chunk c, plane pl, block blk); lane -> slot blk*64 + lane of that plane
    // slot -> (patch row, px, quad parity).  dbase = byte offset of the lane's 16 bytes relative to the patch origin
    // (0x80000000 for slots that hold no pixel: stays out of range after adding any tile origin < 2^31);
    // dpos = px | py << 8 | pz << 16 for the border tiles, whose out-of-image lanes are sent out of range one by one.
    unsigned dbase[NIW];
    int dpos[NIW];
#pragma unroll
    for (int n = 0; n < NIW; ++n) {
        const int i = wave + 4 * n;
        const int c = i / (2 * NBLK), r = i - c * 2 * NBLK, pl = r / NBLK, blk = r - pl * NBLK;
        const int s = blk * 64 + lane;
        const int q1 = s & 1;
        int prow, px;
        if (SW == 1) {
            const int pix = s >> 1;
            prow = pix / PW;
            px = pix - prow * PW;
        } else {
            int t = s >> 1;
            const int xh = t % PWH;
            t /= PWH;
            prow = t >> 1;
            px = 2 * xh + (t & 1);
        }
        const bool valid = i < NI && prow < G::ROWS && px < PW;
        const int pz = KD == 1 ? 0 : prow / PH, py = prow - pz * PH;
        dpos[n] = px | (py << 8) | (pz << 16);
        dbase[n] = valid ? (unsigned)(((pz * a.Hi + py) * a.Wi + px) * (CIN * 4) + (c * 16 + pl * 8 + q1 * 4) * 4) : 0x80000000u;
    }

    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        unsigned q = fast_div(tile, p.tiles_x, p.mul[0], p.shr[0]);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = fast_div(q, p.tiles_y, p.mul[1], p.shr[1]);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = fast_div(q2, (unsigned)a.Do, p.mul[2], p.shr[2]);
        t.zo = (int)(q2 - q3 * (unsigned)a.Do);
        t.b = (int)q3;
        return t;
    };

    auto dma_tile = [&](const TilePos& t, int buf) {
        const int iz0 = t.zo * a.sd - a.pd[0], iy0 = t.ty0 * SW - a.ph[0], ix0 = t.tx0 * SW - a.pw[0];
        // byte offset of the patch origin (may be negative on border tiles: 32-bit wrap-around arithmetic)
        const unsigned origin = (unsigned)((((t.b * a.Di + iz0) * a.Hi + iy0) * a.Wi + ix0) * (CIN * 4));
        const bool inside = iz0 >= 0 && iz0 + KD <= a.Di && iy0 >= 0 && iy0 + PH <= a.Hi && ix0 >= 0 && ix0 + PW <= a.Wi;
        f32x4v* const dst0 = lds + buf * BUF;
        if (inside) {
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const int i = wave + 4 * n;
                if (i < NI) {
                    // (named operands: hipcc 7.2 silently drops the kernel's host stub when this builtin is handed an
                    //  arithmetic expression as its offset inside a generic lambda-free branch like this one)
                    const unsigned off = dbase[n] + origin;
                    f32x4v* const dst = dst0 + (i / NBLK) * PLANE + (i % NBLK) * 64;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const int i = wave + 4 * n;
                if (i < NI) {
                    const int ix = ix0 + (dpos[n] & 255), iy = iy0 + ((dpos[n] >> 8) & 255), iz = iz0 + (dpos[n] >> 16);
                    const bool ok = (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
                    const unsigned off = ok ? dbase[n] + origin : 0x80000000u;
                    f32x4v* const dst = dst0 + (i / NBLK) * PLANE + (i % NBLK) * 64;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                }
            }
        }
    };

    // ---- once per workgroup: weights, scale / shift, A-read bases ------------------------------------------------
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);          // this workgroup's tiles: tile, tile + nwg, ...  (see host side)
    TilePos pos = decode_tile(tile < p.ntiles ? tile : 0);
    if (tile < p.ntiles) dma_tile(pos, 0);
    f32x4v wreg[WREG ? NTAP * NCH * NT : 1];
    const long wstep = (long)a.ntile_total * 256;          // floats per K step of the packed weights
    if (WREG) {
#pragma unroll
        for (int s = 0; s < NTAP * NCH; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wreg[s * NT + nt] = *reinterpret_cast<const f32x4v*>(a.wpk + s * wstep + ((long)(nt0 + nt) * 64 + lane) * 4);
    } else {
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wpk), (short)0, (int)(NTAP * NCH * wstep * 4), 0x00020000);
        for (int i = wave; i < NTAP * NCH * NT; i += 4) {
            const int s = i / NT, nt = i - s * NT;
            const unsigned off = (unsigned)((s * wstep + (long)(nt0 + nt) * 256) * 4) + lane * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wl + i * 64), 16, off, 0, 0, 0);
        }
    }
    f32x4v scv[NT], shv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = (nt0 + nt) * 16 + lq * 4;
        scv[nt] = *reinterpret_cast<const f32x4v*>(a.scale + n0);
        shv[nt] = *reinterpret_cast<const f32x4v*>(a.shift + n0);
    }
    // float4 index (inside a chunk's plane pair) of this lane's A operand for tap (0,0,0) of each of its M tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, x = (t & 1) * 16 + lm;
        const int rs = G::ROWSLOTS;
        abase[mt] = (SW == 1 ? row * rs + x * 2 : row * 2 * rs + x * 2) + (lq >> 1) * PLANE + (lq & 1);
    }
    // output (and skip) byte offset of this lane's 4 channels of each M tile, relative to the tile's first pixel
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.skip_mode == 1 ? a.skip : a.in), (short)0, a.skip_mode == 1 ? (int)p.out_bytes : 0, 0x00020000);
    unsigned obase[MT];
    int orc[MT];            // row | col << 8 of the M tile's pixel inside the workgroup tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, col = (t & 1) * 16 + lm;
        orc[mt] = row | (col << 8);
        obase[mt] = (unsigned)((row * a.Wo + col) * a.cout + nt0 * 16 + lq * 4) * 4u;
    }
    auto tap_off = [&](int kz, int ky, int kx) -> int {     // float4 offset of a tap relative to abase (compile-time)
        const int prow = kz * PH + ky;
        return SW == 1 ? prow * G::ROWSLOTS + kx * 2 : prow * G::ROWSLOTS + (kx & 1) * PWH * 2 + (kx >> 1) * 2;
    };

    __syncthreads();        // (waits vmcnt(0): first patch and the weights have landed)

    for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
        const int cur = it & 1;
        MV_PTL(0);
        MV_PTL(5);
        MV_PTL(6);
        MV_PTL(7);
        const TilePos here = pos;
        // this lane's output offsets (out of range = dropped by the hardware) and the skip values, fetched ahead
        const unsigned oorigin = (unsigned)((((here.b * a.Do + here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
        const bool whole = here.ty0 + TY <= a.Ho && here.tx0 + 32 <= a.Wo;
        unsigned ooff[MT];
        f32x4v skv[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ooff[mt] = obase[mt] + oorigin;
            if (!whole && !(here.ty0 + (orc[mt] & 255) < a.Ho && here.tx0 + (orc[mt] >> 8) < a.Wo)) ooff[mt] = 0x80000000u;
            if (a.skip_mode == 1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    skv[mt][nt] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooff[mt] + nt * 64, 0, 0));
            }
        }
        if (tile + nwg < p.ntiles) {
            pos = decode_tile(tile + nwg);
            dma_tile(pos, cur ^ 1);
        }
        MV_PTL(1);

        f32x4v acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};

        const f32x4v* patch = lds + cur * BUF;
        // depth taps that fall entirely into the zero padding of this output slice are skipped
        const int kz_lo = KD == 1 ? 0 : max(0, a.pd[0] - here.zo * a.sd), kz_hi = KD == 1 ? 1 : min(KD, a.Di + a.pd[0] - here.zo * a.sd);
#pragma unroll
        for (int kz = 0; kz < KD; ++kz) {
            if (kz < kz_lo || kz >= kz_hi) continue;
            // software pipeline over the KH*KW taps of this depth slice: PF taps of operands in flight
            f32x4v A[PF + 1][NCH][MT], Bv[PF + 1][NCH][NT];
            auto load_tap = [&](int t2, f32x4v (&Aa)[NCH][MT], f32x4v (&Bb)[NCH][NT]) {
                const int ky = t2 / KW, kx = t2 - ky * KW;
                const int to = tap_off(kz, ky, kx);
                const int tap = kz * TAPS2D + t2;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) Aa[c][mt] = patch[abase[mt] + c * 2 * PLANE + to];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        Bb[c][nt] = WREG ? wreg[(tap * NCH + c) * NT + nt] : wl[((tap * NCH + c) * NT + nt) * 64 + lane];
                }
            };
#pragma unroll
            for (int t2 = 0; t2 < PF && t2 < TAPS2D; ++t2) load_tap(t2, A[t2 % (PF + 1)], Bv[t2 % (PF + 1)]);
#pragma unroll
            for (int t2 = 0; t2 < TAPS2D; ++t2) {
                if (t2 + PF < TAPS2D) load_tap(t2 + PF, A[(t2 + PF) % (PF + 1)], Bv[(t2 + PF) % (PF + 1)]);
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bv[t2 % (PF + 1)][c][nt][j], A[t2 % (PF + 1)][c][mt][j],
                                                                                   acc[mt][nt], 0, 0, 0);
            }
        }

        MV_PTL(2);
        __syncthreads();    // vmcnt(0): the next tile's patch has landed; everyone is done reading this one
        MV_PTL(3);

        // epilogue: the accumulator is D^T (weights in the A slot): 4 consecutive output channels of one voxel per lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4v v = acc[mt][nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = fmaf(v[j], scv[nt][j], shv[nt][j]);
                    if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                    if (a.skip_mode == 1) v[j] += skv[mt][nt][j];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, ooff[mt] + nt * 64, 0, 0);
            }
        }
        MV_PTL(4);
    }
}

int g_num_cu = 0;

template <int MT, int NT, int KW, int SW, int NCH, int KD, bool WREG, int PF>
int launch_pers(const ConvArgs& a, int wpc, hipStream_t s) {
    using G = PersGeom<MT, KW, SW, KD>;
    constexpr int NTAP = KD * KW * KW;
    const size_t lds = (size_t)(2 * NCH * 2 * G::PLANE + (WREG ? 0 : NTAP * NCH * NT * 64)) * 16;
    if (lds > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv_pers_kernel<MT, NT, KW, SW, NCH, KD, WREG, PF>;
    static bool attr_set = false;
    if (!attr_set) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MVSTER_ERR_LAUNCH;
        attr_set = true;
    }
    if (g_num_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MVSTER_ERR_LAUNCH;
        g_num_cu = prop.multiProcessorCount;
    }
    PersArgs p;
    p.tiles_x = (unsigned)((a.Wo + 31) / 32);
    p.tiles_y = (unsigned)((a.Ho + G::TY - 1) / G::TY);
    const long ntiles = (long)p.tiles_x * p.tiles_y * a.Do * a.B;
    // 32-bit byte offsets, and 0x80000000 + any in-tensor offset must stay out of range: both tensors < 2 GB
    const long out_bytes = (long)a.B * a.DoF * a.HoF * a.WoF * a.cout * 4;
    if (ntiles >= (1L << 30) || a.in_bytes >= (1u << 31) || out_bytes >= (1L << 31)) return MVSTER_ERR_UNSUPPORTED;
    p.ntiles = (unsigned)ntiles;
    p.out_bytes = (unsigned)out_bytes;
    const unsigned divisors[3] = {p.tiles_x, p.tiles_y, (unsigned)a.Do};
    for (int i = 0; i < 3; ++i) find_divisor(divisors[i], p.mul[i], p.shr[i]);
    // resident workgroups per CU: what LDS allows, at most 4 (registers), unless the caller pins it
    int by_lds = (int)((160 * 1024) / lds);
    int per_cu = wpc > 0 ? wpc : (by_lds > 3 ? 3 : by_lds);
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    const int ny = a.ntile_total / NT;
    long gmax = (long)g_num_cu * per_cu / ny;
    if (gmax < 1) gmax = 1;
    // equal shares: every workgroup walks the same number of tiles (no straggler round)
    const long rounds = (ntiles + gmax - 1) / gmax;
    const long gx = (ntiles + rounds - 1) / rounds;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, ny, 1), dim3(256), lds, s, a, p);
    return mv_check_launch();
}

}  // namespace

#ifdef MVSTER_TIMELINE
extern "C" int mvster_debug_pers_timeline(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ptl), &buf, sizeof(buf)) == hipSuccess ? MVSTER_OK : MVSTER_ERR_LAUNCH;
}
#endif

// Layers the family covers: ordinary (non-transposed) convolutions, cin in {16, 32}, cout % 16 == 0, kernel (1|3) x 3 x 3
// or 1 x 5 x 5 with "same" padding geometry handled by the generic bounds checks, stride 1 or 2 in-plane.
int dispatch_pers(const ConvArgs& a, int mt, int nt, int wpc, hipStream_t s) {
    if (a.nclass != 1 || a.osd != 1 || a.osh != 1 || a.osw != 1 || a.skip_mode > 1 || a.prob_w || a.cout % 16 != 0 ||
        a.sh != a.sw || a.kh[0] != a.kw[0] || a.ntile_total % nt != 0 || mt != 2)
        return MVSTER_ERR_UNSUPPORTED;
    const int kd = a.kd[0], kw = a.kw[0], sw = a.sw, nch = a.cin / 16;
    if (a.cin % 16 != 0) return MVSTER_ERR_UNSUPPORTED;
#define MV_P(NT_, KW_, SW_, NCH_, KD_, WREG_, PF_) \
    if (nt == NT_ && kw == KW_ && sw == SW_ && nch == NCH_ && kd == KD_) return launch_pers<2, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_>(a, wpc, s);
    MV_P(1, 3, 1, 1, 1, true, 2)      // 16 -> 16 3x3           (FPN conv1.1/1.2, composed mid level)
    MV_P(2, 3, 1, 2, 1, false, 1)     // 32 -> 32 3x3           (FPN conv2.1/2.2)
    MV_P(2, 5, 2, 1, 1, false, 1)     // 16 -> 32 5x5 stride 2  (FPN conv2.0)
    MV_P(1, 3, 1, 1, 3, false, 2)     // 16 -> 16 3x3x3         (reg2d conv2)
    MV_P(2, 3, 2, 1, 1, false, 1)     // 16 -> 32 3x3 stride 2  (reg2d conv3)
#undef MV_P
    return MVSTER_ERR_UNSUPPORTED;
}

}  // namespace mvconv
