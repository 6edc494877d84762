// Winograd F(2x2, 3x3) form of the persistent LDS-DMA convolution (variant 8 of mvster_conv_mfma).
//
// Why: the 3x3 stride-1 layers of FPN4 / reg2d are bound by the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: 32 cycles per
// 1 024 MACs and SIMD), and conv_pers_kernel already spends 60-75 % of its tile time inside MFMAs.
// The minimal-filtering form  Y = A^T [ (G g G^T) . (B^T d B) ] A  (Lavin & Gray, "Fast Algorithms for Convolutional
// Neural Networks", 2015) computes a 2x2 output block from a 4x4 input block with 16 multiplications per (cin, cout)
// instead of 36: 2.25x fewer MFMAs, paid with ~56 VALU additions per lane and 16-channel chunk.  The element-wise product
// over the 16 transform points is 16 independent [tiles x cin] x [cin x cout] GEMMs -- MFMA work with K = cin.
//
//   * input side: same LDS patch (and the same LDS-DMA address decode) as conv_pers_kernel, TY = 8 rows x 32 pixels per
//     workgroup tile.  Wave w owns the 16 Winograd blocks of output rows 2w, 2w+1 (lane & 15 = block column); a lane
//     reads its 4x4 pixels x 4 channels (16 ds_read_b128 per 16-channel chunk), transforms them in registers (B^T d B has
//     only 0 / +-1 coefficients: 32 float4 additions) and feeds the 16 results as MFMA operands;
//   * weights: U = G g G^T is computed when the layer is packed (mvster_pack_wino_weights), stored in the packed
//     fragment order with the 16 transform points in place of the 9 taps; in registers for 16-channel inputs, else LDS;
//   * output side: a lane ends up with all 16 points of ITS block and 4 output channels, so A^T M A (24 float4
//     additions) and the fused epilogue (scale/shift, ReLU, same-shape skip) need no exchange; four float4 stores.
//
// Not bit-identical to the direct kernels (different operation order and the 1/2, 1/4 factors folded into U); fp32
// throughout, error a few ulp of the accumulated magnitude (tests/test_gpu_kernels.py bounds it against the direct kernel
// and the fp64 oracle).
//
// Reference layers: Conv2d(3x3) + BatchNorm + ReLU of FPN4 (models/mvs4net_utils.py:419-502) and the (1,3,3) / per-slice
// taps of ConvBnReLU3D in reg2d (:870-912).
#include "conv_args.hpp"

namespace mvconv {
namespace {

typedef float f32x2v __attribute__((ext_vector_type(2)));

// Probe build only (make timeline; scripts/conv_wino_timeline.py): s_memtime stamps of conv_wino_ring_kernel, per workgroup,
// wave and step (< 32): [0] top of the step, [1] MFMAs issued, [2] transform done (before the end-of-step wait), [3] past
// the barrier.
#ifdef MVSTER_TIMELINE
__device__ unsigned long long* g_wtl = nullptr;
#define MV_WTL(k)                                                                                                       \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (g_wtl && lane == 0 && g < 32)                                                                               \
            g_wtl[(((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave8) * 32 + g) * 4 + (k)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define MV_WTL(k)
#endif

// Packed fp32 additions: one instruction per TWO lanes-elements (the transforms are additions only; hipcc emits v_pk_add_f32
// for a + b but four v_sub_f32 for a float4 subtraction -- the negation is an operand modifier of the packed form).
// The hazard recogniser does not look inside inline assembly: results of MFMAs reach these only behind the explicit
// s_nop block after the MFMA phase (see the main loop).
#ifndef MV_WINO_SCALAR_ADDS
#define MV_WINO_SCALAR_ADDS 0
#endif
#if MV_WINO_SCALAR_ADDS
// two plain additions per register pair: a wave64 v_add_f32 issues in 2 cycles on gfx950, v_pk_add_f32 was measured at 8
__device__ __forceinline__ f32x2v pk_add(f32x2v a, f32x2v b) {
    float r0, r1;
    asm("v_add_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(r0), "=&v"(r1) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]));
    return (f32x2v){r0, r1};
}
__device__ __forceinline__ f32x2v pk_sub(f32x2v a, f32x2v b) {
    float r0, r1;
    asm("v_sub_f32 %0, %2, %3\n\tv_sub_f32 %1, %4, %5" : "=&v"(r0), "=&v"(r1) : "v"(a[0]), "v"(b[0]), "v"(a[1]), "v"(b[1]));
    return (f32x2v){r0, r1};
}
#else
__device__ __forceinline__ f32x2v pk_add(f32x2v a, f32x2v b) {
    f32x2v r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2v pk_sub(f32x2v a, f32x2v b) {
    f32x2v r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif
struct Q4 {                // a float4 kept as two register pairs
    f32x2v lo, hi;
};
__device__ __forceinline__ Q4 operator+(const Q4& a, const Q4& b) { return {pk_add(a.lo, b.lo), pk_add(a.hi, b.hi)}; }
__device__ __forceinline__ Q4 operator-(const Q4& a, const Q4& b) { return {pk_sub(a.lo, b.lo), pk_sub(a.hi, b.hi)}; }
__device__ __forceinline__ float elem(const Q4& a, int j) { return j < 2 ? a.lo[j] : a.hi[j - 2]; }

// NT: N tiles (16 output channels) per workgroup; NCH = cin / 16; WREG: U in registers (16*NCH*NT float4 per lane);
// SKIP: a same-shape tensor is added in the epilogue.
template <int NT, int NCH, bool WREG, bool SKIP>
__global__ void __launch_bounds__(256) conv_wino_kernel(ConvArgs a, PersArgs p) {
    using G = PersGeom<4, 3, 1, 1>;
    constexpr int TY = G::TY, PW = G::PW, PLANE = G::PLANE, NBLK = G::NBLK, RS = G::ROWSLOTS;
    constexpr int BUF = NCH * 2 * PLANE;                    // float4 per patch buffer
    constexpr int NI = NCH * 2 * NBLK;                      // DMA wave-instructions per tile
    constexpr int NIW = (NI + 3) / 4;
    constexpr int CIN = NCH * 16;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const scratch = lds + 2 * BUF;                  // 64 float4: target of the surplus DMA slots (NI % 4 != 0)
    f32x4v* const wl = scratch + 64;                        // [point][chunk][nt][lane]  (unused with WREG)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.y * NT;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);

    // ---- LDS-DMA address decode: as conv_pers_kernel (stride 1, one depth slice) -----------------------------------
    unsigned dbase[NIW];
    int dpos[NIW];
#pragma unroll
    for (int n = 0; n < NIW; ++n) {
        const int i = wave + 4 * n;
        const int c = i / (2 * NBLK), r = i - c * 2 * NBLK, pl = r / NBLK, blk = r - pl * NBLK;
        const int s = blk * 64 + lane;
        const int q1 = s & 1, pix = s >> 1;
        // a patch row holds its even columns first, then the odd ones (as the ring kernel below: a lane's 4x4 block starts
        // at column 2*lm, so the 16 lanes of a ds_read_b128 are 32 bytes apart instead of 64 -- no 2-way bank conflict)
        const int py = pix / PW, ps = pix - py * PW;
        const int px = ps < G::PWH ? 2 * ps : 2 * (ps - G::PWH) + 1;
        const bool valid = i < NI && py < G::ROWS;
        dpos[n] = px | (py << 8);
        dbase[n] = valid ? (unsigned)((py * a.Wi + px) * (CIN * 4) + (c * 16 + pl * 8 + q1 * 4) * 4) : 0x80000000u;
    }
    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        auto div = [&](unsigned n, int k) -> unsigned { return ((__umulhi(n, p.mul[k]) >> p.shr[k]) & ~p.one[k]) | (n & p.one[k]); };
        unsigned q = div(tile, 0);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = div(q, 1);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = div(q2, 2);
        t.zo = (int)(q2 - q3 * (unsigned)a.Do);
        t.b = (int)q3;
        return t;
    };
    auto dma_tile = [&](const TilePos& t, int buf, bool live) {
        const int iy0 = t.ty0 - a.ph[0], ix0 = t.tx0 - a.pw[0];
        const unsigned origin = (unsigned)((((t.b * a.Di + t.zo) * a.Hi + iy0) * a.Wi + ix0) * (CIN * 4));
        const unsigned wi = live ? (unsigned)a.Wi : 0u;
        f32x4v* const dst0 = lds + buf * BUF;
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int i = wave + 4 * n;
            const int ix = ix0 + (dpos[n] & 255), iy = iy0 + (dpos[n] >> 8);
            const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < wi;
            // (named operands: hipcc 7.2 silently drops the kernel's host stub when this builtin is handed an arithmetic
            //  expression as its offset)
            const unsigned off = ok ? dbase[n] + origin : 0x80000000u;
            f32x4v* const dst = (NI % 4 == 0 || n + 1 < NIW || i < NI) ? dst0 + (i / NBLK) * PLANE + (i % NBLK) * 64 : scratch;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
    };

    // ---- once per workgroup -------------------------------------------------------------------------------------------
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);
    TilePos pos = decode_tile(tile < p.ntiles ? tile : 0);
    if (tile < p.ntiles) dma_tile(pos, 0, true);
    f32x4v wreg[WREG ? 16 * NCH * NT : 1];
    const long wstep = (long)a.ntile_total * 256;          // floats per K step of the packed weights
    if (WREG) {
#pragma unroll
        for (int s = 0; s < 16 * NCH; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wreg[s * NT + nt] = *reinterpret_cast<const f32x4v*>(a.wpk + s * wstep + ((long)(nt0 + nt) * 64 + lane) * 4);
    } else {
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wpk), (short)0, (int)(16 * NCH * wstep * 4), 0x00020000);
        for (int i = wave; i < 16 * NCH * NT; i += 4) {
            const int s = i / NT, nt = i - s * NT;
            const unsigned off = (unsigned)((s * wstep + (long)(nt0 + nt) * 256) * 4) + lane * 16;
            f32x4v* const dst = wl + i * 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
    }
    f32x4v scv[NT], shv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = (nt0 + nt) * 16 + lq * 4;
        scv[nt] = *reinterpret_cast<const f32x4v*>(a.scale + n0);
        shv[nt] = *reinterpret_cast<const f32x4v*>(a.shift + n0);
    }
    // float4 index of this lane's 4x4 block origin: patch row 2*wave, column 2*lm (even-column slot lm), its channel quad
    const int abase = 2 * wave * RS + 2 * lm + (lq >> 1) * PLANE + (lq & 1);
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)p.out_bytes : 0, 0x00020000);
    // byte offset of output pixel (2*wave, 2*lm), this lane's 4 channels, relative to the tile's first pixel
    const unsigned obase = (unsigned)((2 * wave * a.Wo + 2 * lm) * a.cout + nt0 * 16 + lq * 4) * 4u;
    const unsigned opix = (unsigned)a.cout * 4u, orow = (unsigned)a.Wo * opix;

    __syncthreads();        // (first patch and the weights have landed)

    for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
        const int cur = it & 1;
        const TilePos here = pos;
        const unsigned oorigin = (unsigned)((((here.b * a.Do + here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
        unsigned ooff[2][2];
        f32x4v skv[2][2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = here.ty0 + 2 * wave + i < a.Ho && here.tx0 + 2 * lm + j < a.Wo;
                ooff[i][j] = ok ? obase + oorigin + i * orow + j * opix : 0x80000000u;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    skv[i][j][nt] = SKIP ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooff[i][j] + nt * 64, 0, 0))
                                         : (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
        const bool has_next = tile + nwg < p.ntiles;
        pos = decode_tile(has_next ? tile + nwg : tile);
        dma_tile(pos, cur ^ 1, has_next);
        __builtin_amdgcn_sched_barrier(0);

        f32x4v acc[16][NT];
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[q][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        const f32x4v* patch = lds + cur * BUF + abase;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
            Q4 d[4][4], V[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const f32x4v v = patch[c * 2 * PLANE + r * RS + (x & 1) * (G::PWH * 2) + (x >> 1) * 2];
                    d[r][x] = {{v[0], v[1]}, {v[2], v[3]}};
                }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const Q4 t0 = d[0][x] - d[2][x], t1 = d[1][x] + d[2][x], t2 = d[2][x] - d[1][x], t3 = d[1][x] - d[3][x];
                d[0][x] = t0;
                d[1][x] = t1;
                d[2][x] = t2;
                d[3][x] = t3;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[r][0] = d[r][0] - d[r][2];
                V[r][1] = d[r][1] + d[r][2];
                V[r][2] = d[r][2] - d[r][1];
                V[r][3] = d[r][1] - d[r][3];
            }
            // 16 independent accumulators per N tile: consecutive MFMAs never depend on each other
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 3" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f32x4v u = WREG ? wreg[(q * NCH + c) * NT + nt] : wl[((q * NCH + c) * NT + nt) * 64 + lane];
                        acc[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j], elem(V[q >> 2][q & 3], j), acc[q][nt], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (the packed additions below are inline assembly: the MFMA results they read are separated by hand -- 24 wait
        //  states cover the 8-pass MFMA's write-back, ISA "XDL write VGPR -> VALU read")
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; then the fused epilogue on the 2x2 pixels x 4 channels of this lane
        const float floor_v = a.relu ? 0.0f : -__builtin_inff();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            Q4 m[16], t[2][4];
#pragma unroll
            for (int q = 0; q < 16; ++q) m[q] = {{acc[q][nt][0], acc[q][nt][1]}, {acc[q][nt][2], acc[q][nt][3]}};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                t[0][x] = m[0 + x] + m[4 + x] + m[8 + x];
                t[1][x] = m[4 + x] - (m[8 + x] + m[12 + x]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                Q4 y[2];
                y[0] = t[i][0] + t[i][1] + t[i][2];
                y[1] = t[i][1] - (t[i][2] + t[i][3]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4v v = {y[j].lo[0], y[j].lo[1], y[j].hi[0], y[j].hi[1]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaxf(fmaf(v[e], scv[nt][e], shv[nt][e]), floor_v);
                        if (SKIP) v[e] += skv[i][j][nt][e];
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, ooff[i][j] + nt * 64, 0, MV_STORE_AUX);
                }
            }
        }
        // the next tile's patch has landed once at most this tile's stores (issued after its DMA) are outstanding; then
        // everyone is done reading this tile's patch
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NT) : "memory");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    }
}

template <int NT, int NCH, bool WREG, bool SKIP>
int launch_wino(const ConvArgs& a, int wpc, hipStream_t s) {
    using G = PersGeom<4, 3, 1, 1>;
    const size_t lds = (size_t)(2 * NCH * 2 * G::PLANE + 64 + (WREG ? 0 : 16 * NCH * NT * 64)) * 16;
    if (lds > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv_wino_kernel<NT, NCH, WREG, SKIP>;
    static unsigned long attr_done = 0;
    if (lds > 64 * 1024 && !allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    PersArgs p;
    if (!fill_pers_args(a, G::TY, p)) return MVSTER_ERR_UNSUPPORTED;
    const long ntiles = p.ntiles;
    const int by_lds = (int)((160 * 1024) / lds);
    int per_cu = wpc > 0 ? wpc : 2;
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    const int ny = a.ntile_total / NT;
    long gmax = (long)ncu * per_cu / ny;
    if (gmax < 1) gmax = 1;
    const long rounds = (ntiles + gmax - 1) / gmax;       // equal shares: every workgroup walks the same number of tiles
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv_wino_kernel<%d, %d, %s, %s>", NT, NCH, WREG ? "true" : "false", SKIP ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, ny, 1), dim3(256), lds, s, a, p);
    return mv_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// The same transform for the deep layers: 3x3x3 kernels (reg2d's conv2/4/6, models/mvs4net_utils.py:885-891) and 64-channel
// 3x3 layers, whose patches and transformed weights do not fit LDS as a whole.  The K dimension is walked in STEPS of one
// depth tap and one 16-channel chunk: a step's input is one patch slice (10 x 34 pixels x 16 channels, 22 KB), its weights
// U[kz][.][chunk] are 16 KB per N tile.  Both stream through LDS rings filled by LDS-DMA:
//   * patch ring of 4 slices: the slice of step g+3 is requested at the top of step g, so that at the start of step g the
//     slices of g and g+1 are known to have landed -- the 16 ds_read_b128 of step g+1 are issued BEFORE the MFMAs of step
//     g and its transform runs after them (one wave per SIMD: nothing else hides that latency);
//   * weight ring of 2: U of step g+1 is requested at the top of step g (issued before the patch request, so the counted
//     s_waitcnt at the end of the step covers it), read back in groups of 4 transform points, one group ahead of the MFMAs;
//   * one barrier per step; the 16*NT accumulators live across the steps of a tile, the output transform and the fused
//     epilogue run at its last step.  Depth taps that fall into the zero padding are skipped (no step).
// Patch rows are stored with even and odd columns apart (a lane's 4x4 block starts at column 2*lm: the 16 lanes of a read
// are then 32 bytes apart -- the 64-byte stride of the plain layout costs a 2-way bank conflict on every read,
// SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE in profiles/r03_k_wino_pmc.txt).
// ------------------------------------------------------------------------------------------------------------------
struct RingGeom {
    static constexpr int TY = 8, PWH = 17, PH = 10;                     // 34 patch columns as 17 even + 17 odd
    static constexpr int ROWSLOTS = PWH * 4;                              // float4 slots of one patch row in one plane
    static constexpr int USED = PH * ROWSLOTS;
    static constexpr int NBLK = (USED + 63) / 64;                         // DMA wave-instructions per plane
    static constexpr int PLANE = ((NBLK * 64 + 7) & ~7) + 4;              // plane pitch (float4)
    static constexpr int SLICE = 2 * PLANE;                               // float4 per ring slot
};

struct RingStep {          // one step of one tile: everything wave-uniform
    unsigned tile;
    int kz, c, kz_hi;
    TilePos pos;
    bool live;
};

// Eight waves per workgroup, two per SIMD: waves 0-3 are the four tile rows of one N tile; waves 4-7 either compute the
// SECOND N tile of the same pixels (SPLITN: the layer has >= 32 output channels; both halves read the same slices, each
// issues half of the DMA) or only issue the DMA (cout == 16).  Why: an LDS-DMA instruction stalls the issuing wave for
// 60-180 cycles (MI355X_MICROARCH "LDS-DMA piece issue cost"), ~10 of them per step; with one wave per SIMD the matrix pipe
// idles through every one of those stalls (first version, 4 waves: 4 000 cycles per 64-MFMA step), with a second wave on the
// SIMD the pipe has other MFMAs to run.  (VALU work is not hidden that way: fp32 MFMAs and VALU instructions of two waves on
// one SIMD were never seen to overlap -- conv_pp_kernel in conv_pers.hip is the experiment.)
// MODE 0: one N tile, waves 4-7 load.  MODE 1 (SPLITN): waves 4-7 compute the second N tile, every wave loads.  MODE 2: the
// compute waves hold BOTH N tiles (one input transform for 128 MFMAs instead of 64; no read-ahead of the next step's pixels:
// the registers are gone) and waves 4-7 load.
template <int NCH, int KD, bool SKIP, int MODE>
__global__ void __launch_bounds__(512) conv_wino_ring_kernel(ConvArgs a, PersArgs p) {
    constexpr bool SPLITN = MODE == 1;
    constexpr int NT = MODE == 2 ? 2 : 1;                   // N tiles per compute wave
    constexpr int NTW = MODE == 0 ? 1 : 2;                  // N tiles per workgroup
    constexpr bool AHEAD = NT == 1;                         // next step's pixels are read under this step's MFMAs
    using G = RingGeom;
    constexpr int TY = G::TY, PLANE = G::PLANE, NBLK = G::NBLK, RS = G::ROWSLOTS, SLICE = G::SLICE;
    constexpr int NLW = SPLITN ? 8 : 4;                     // waves that issue DMA
    constexpr int NIW = (NBLK + NLW / 2 - 1) / (NLW / 2);  // slice DMA instructions per loading wave and step (one plane's share)
    constexpr int NUW = 16 * NTW / NLW;                     // weight DMA instructions per loading wave and step
    constexpr int CIN = NCH * 16;
    constexpr int UST = 16 * NTW * 64;                      // float4 per weight ring slot
    constexpr bool URES = KD * NCH * NTW <= 4;              // all transformed weights of the workgroup fit 64 KB: fetched once
    constexpr int USLOTS = URES ? KD * NCH : 2;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const ring = reinterpret_cast<f32x4v*>(lds_raw);              // 4 patch slices
    f32x4v* const uring = ring + 4 * SLICE;                               // 2 weight slots [point][nt][lane]
    f32x4v* const scratch = uring + USLOTS * UST;                         // 64 float4: surplus DMA slots

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave8 >> 2, wave = wave8 & 3;          // wave: tile row pair (compute) 
    const bool loads = SPLITN || half == 1;
    const int lw = SPLITN ? wave8 : wave;                   // index among the loading waves
    const int lm = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.y * NTW;                       // first N tile of the workgroup
    const int ntw = SPLITN ? half : 0;                      // this wave's N tile inside the workgroup
    // Consecutive LDS-DMA instructions of a wave that target consecutive kilobytes of LDS share ONE M0 value and differ in
    // the instruction's immediate offset (0, 1024, 2048, 3072) -- rewriting M0 between two of them makes the second wait
    // until the first has landed (measured: 270-350 cycles per instruction, profiles/r03_k_wino_timeline.txt).  The
    // immediate also moves the global address, so instruction k reads through a descriptor whose base is 1024*k lower.
    const long wstep = (long)a.ntile_total * 256;          // floats per K step (16 channels) of the packed weights
    auto in_rsrc = [&](int k) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in) - 256 * k, (short)0, (int)(a.in_bytes + 1024u * k), 0x00020000);
    };
    auto w_rsrc = [&](int k) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk) - 256 * k, (short)0,
                                                 (int)(KD * 16 * NCH * wstep * 4 + 1024 * k), 0x00020000);
    };

    // ---- LDS-DMA address decode of one slice: instruction i = lw + NLW*n -> (plane, block); lane -> slot -> (row, px, quad)
    // dbase = byte offset of the lane's 16 bytes relative to the slice origin (0x80000000: the slot holds no pixel),
    // dpos = px | row << 8 for the border test.  Kept in registers when every wave loads (3 pieces each); recomputed per
    // step by the loading waves of the cout == 16 form (6 pieces: 12 registers the compute waves cannot spare).
    static_assert((NLW / 2) * NIW >= NBLK, "the waves of one plane cover its blocks");
    const int lpl = lw / (NLW / 2), lblk0 = (lw % (NLW / 2)) * NIW;       // this wave's plane and first block
    auto decode_piece = [&](int n, int ln, unsigned& db, int& dp) {
        const int pl = lpl, blk = lblk0 + n;
        const int s = blk * 64 + ln;
        const int q1 = s & 1;
        int t = s >> 1;
        const int xh = t % G::PWH;
        t /= G::PWH;
        const int py = t >> 1, px = 2 * xh + (t & 1);
        const bool valid = blk < NBLK && py < G::PH;
        dp = px | (py << 8);
        db = valid ? (unsigned)((py * a.Wi + px) * (CIN * 4) + (pl * 8 + q1 * 4) * 4) : 0x80000000u;
    };
    unsigned dbase[SPLITN ? NIW : 1];
    int dpos[SPLITN ? NIW : 1];
    if (SPLITN) {
#pragma unroll
        for (int n = 0; n < NIW; ++n) decode_piece(n, lane, dbase[n], dpos[n]);
    }
    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        auto div = [&](unsigned n, int k) -> unsigned { return ((__umulhi(n, p.mul[k]) >> p.shr[k]) & ~p.one[k]) | (n & p.one[k]); };
        unsigned q = div(tile, 0);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = div(q, 1);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = div(q2, 2);
        t.zo = (int)(q2 - q3 * (unsigned)a.Do);
        t.b = (int)q3;
        return t;
    };
    const unsigned nwg = gridDim.x;
    auto first_step = [&](unsigned tile) -> RingStep {
        RingStep st;
        st.tile = tile;
        st.live = tile < p.ntiles;
        st.pos = decode_tile(st.live ? tile : 0u);
        st.kz = KD == 1 ? 0 : max(0, a.pd[0] - st.pos.zo);                 // depth taps inside the volume (stride 1)
        st.kz_hi = KD == 1 ? 1 : min(KD, a.Di + a.pd[0] - st.pos.zo);
        st.c = 0;
        return st;
    };
    auto is_last = [&](const RingStep& st) -> bool { return st.c == NCH - 1 && st.kz == st.kz_hi - 1; };
    auto next_step = [&](const RingStep& st) -> RingStep {
        if (!st.live) return st;
        if (is_last(st)) return first_step(st.tile + nwg);
        RingStep n = st;
        if (++n.c == NCH) {
            n.c = 0;
            ++n.kz;
        }
        return n;
    };
    // One slice / weight request = NIW / NUW LDS-DMA instructions of this wave, issued one by one ("pieces") so that the
    // compute waves can spread them over the MFMA phase: an LDS-DMA instruction holds its wave for 60-180 cycles, which the
    // other wave of the SIMD fills with its MFMAs only if that wave is not stalled at the same place.
    // Interior slices (all 10 x 34 pixels inside the image) need no per-lane test: the lane's offset is a constant and
    // the slice origin goes into the instruction's scalar offset -- no VALU work at all.
    struct SliceReq {
        unsigned origin, wi;
        int iy0, ix0;
        bool interior;
        f32x4v* dst0;
    };
    auto slice_req = [&](const RingStep& st, int slot) -> SliceReq {
        SliceReq r;
        const int iz = st.pos.zo + st.kz - a.pd[0];
        r.iy0 = st.pos.ty0 - a.ph[0];
        r.ix0 = st.pos.tx0 - a.pw[0];
        r.origin = (unsigned)(((((st.pos.b * a.Di + iz) * a.Hi + r.iy0) * a.Wi + r.ix0) * CIN + st.c * 16) * 4);
        r.wi = st.live ? (unsigned)a.Wi : 0u;
        r.interior = st.live && r.iy0 >= 0 && r.ix0 >= 0 && r.iy0 + G::PH <= a.Hi && r.ix0 + 2 * G::PWH <= a.Wi;
        r.dst0 = ring + slot * SLICE;
        return r;
    };
    auto slice_piece = [&](const SliceReq& r, int n, unsigned db, int dp) {
        // pieces 4k .. 4k+3 of this wave: M0 = the wave's first block + 4 KB * k, immediate 1024 * (n % 4); the piece past
        // the plane's last block (one wave per plane has it) goes to the scratch block, every lane out of range
        constexpr int IMM[4] = {0, 1024, 2048, 3072};
        const bool real = lblk0 + n < NBLK;
        f32x4v* const dst = (NIW * (NLW / 2) == NBLK || n + 1 < NIW || real) ? r.dst0 + lpl * PLANE + (lblk0 + (n & ~3)) * 64
                                                                             : scratch - (n & 3) * 64;
        // (named operands: hipcc 7.2 silently drops the kernel's host stub when this builtin is handed an arithmetic
        //  expression as its offset)
        const __amdgpu_buffer_rsrc_t rs = in_rsrc(n & 3);
        if (r.interior) {
            const unsigned so = r.origin;
            switch (n & 3) {
                case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[0], 0); break;
                case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[1], 0); break;
                case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[2], 0); break;
                default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[3], 0); break;
            }
        } else {
            const int ix = r.ix0 + (dp & 255), iy = r.iy0 + (dp >> 8);
            const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < r.wi;
            const unsigned off = ok ? db + r.origin : 0x80000000u;
            switch (n & 3) {
                case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[0], 0); break;
                case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[1], 0); break;
                case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[2], 0); break;
                default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[3], 0); break;
            }
        }
    };
    struct WeightReq {
        unsigned base;
        f32x4v* dst0;
    };
    auto weight_req = [&](const RingStep& st, int slot) -> WeightReq {
        // K steps of this (kz, chunk): (kz*16 + point)*NCH + c; a dead step asks beyond the array (nothing is fetched)
        WeightReq r;
        r.base = st.live ? (unsigned)(((long)(st.kz * 16 * NCH + st.c) * wstep + (long)nt0 * 256) * 4) : 0x80000000u;
        r.dst0 = uring + slot * UST;
        return r;
    };
    const unsigned wlane = lane * 16;
    static_assert(NUW <= 8, "two M0 values per weight request");
    auto weight_piece = [&](const WeightReq& r, int m) {
        const int u = lw * NUW + m;                         // -> (point, nt); this wave's NUW kilobytes are consecutive
        const int q = u / NTW, nt = u - q * NTW;
        const unsigned so = r.base + (unsigned)(((long)q * NCH * wstep + nt * 256) * 4);
        f32x4v* const dst = r.dst0 + (lw * NUW + (m & ~3)) * 64;
        const __amdgpu_buffer_rsrc_t rs = w_rsrc(m & 3);
        switch (m & 3) {
            case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 0, 0); break;
            case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 1024, 0); break;
            case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 2048, 0); break;
            default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 3072, 0); break;
        }
    };

    // ---- once per workgroup -------------------------------------------------------------------------------------------
    f32x4v scv[NT], shv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = (nt0 + ntw + nt) * 16 + lq * 4;
        scv[nt] = *reinterpret_cast<const f32x4v*>(a.scale + n0);
        shv[nt] = *reinterpret_cast<const f32x4v*>(a.shift + n0);
    }
    // float4 index of this lane's 4x4 block origin inside a slice: row 2*wave, column pair lm, its channel quad
    const int abase = 2 * wave * RS + 2 * lm + (lq >> 1) * PLANE + (lq & 1);
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)p.out_bytes : 0, 0x00020000);
    const unsigned obase = (unsigned)((2 * wave * a.Wo + 2 * lm) * a.cout + (nt0 + ntw) * 16 + lq * 4) * 4u;
    const unsigned opix = (unsigned)a.cout * 4u, orow = (unsigned)a.Wo * opix;
    const float floor_v = a.relu ? 0.0f : -__builtin_inff();

    Q4 d[4][4], V[4][4];
    auto read_block = [&](int slot) {
        const f32x4v* patch = ring + slot * SLICE + abase;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const f32x4v v = patch[r * RS + (x & 1) * (G::PWH * 2) + (x >> 1) * 2];
                d[r][x] = {{v[0], v[1]}, {v[2], v[3]}};
            }
    };
    auto transform_block = [&]() {       // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const Q4 t0 = d[0][x] - d[2][x], t1 = d[1][x] + d[2][x], t2 = d[2][x] - d[1][x], t3 = d[1][x] - d[3][x];
            d[0][x] = t0;
            d[1][x] = t1;
            d[2][x] = t2;
            d[3][x] = t3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            V[r][0] = d[r][0] - d[r][2];
            V[r][1] = d[r][1] + d[r][2];
            V[r][2] = d[r][2] - d[r][1];
            V[r][3] = d[r][1] - d[r][3];
        }
    };

    // steps in flight: s0 = the one computed now, s1 = next (its weights are requested now), s3 = three ahead (its slice is)
    RingStep s0 = first_step(xcd_remap(blockIdx.x, nwg));
    RingStep s1 = next_step(s0), s2 = next_step(s1), s3 = next_step(s2);
    if (loads) {
        // (the loading waves of the cout == 16 form keep their own copy of the decode: it is live only on their path)
        unsigned db0[NIW];
        int dp0[NIW];
#pragma unroll
        for (int n = 0; n < NIW; ++n) decode_piece(n, lane, db0[n], dp0[n]);
        const RingStep* const first[3] = {&s0, &s1, &s2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const SliceReq r = slice_req(*first[k], k);
#pragma unroll
            for (int n = 0; n < NIW; ++n) slice_piece(r, n, db0[n], dp0[n]);
        }
        if (URES) {
            // resident weights: slot kz * NCH + c, once per workgroup (a step's 16 KB per N tile would otherwise be
            // re-fetched at every step: with them the CU asks for 38 KB per 64-MFMA step, more than it gets from L2 / MALL
            // in that time -- scripts/probes/lds_dma_bw.hip, profiles/r03_k_lds_dma_bw.txt)
#pragma unroll
            for (int kz = 0; kz < KD; ++kz)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    RingStep st = s0;
                    st.live = true;
                    st.kz = kz;
                    st.c = c;
                    const WeightReq wq = weight_req(st, kz * NCH + c);
#pragma unroll
                    for (int m = 0; m < NUW; ++m) weight_piece(wq, m);
                }
        } else {
            const WeightReq w0 = weight_req(s0, 0);
#pragma unroll
            for (int m = 0; m < NUW; ++m) weight_piece(w0, m);
        }
    }
    __syncthreads();        // (waits for everything requested so far)
    if (!SPLITN && half == 1) {
        // ---- loading waves (cout == 16 form): a loop of their own -- same barrier count, none of the compute registers.
        // Per step: this step's requests, then wait until slice g+2 and weights g+1 have landed.
        unsigned dbl[NIW];
        int dpl[NIW];
#pragma unroll
        for (int n = 0; n < NIW; ++n) decode_piece(n, lane, dbl[n], dpl[n]);
        for (int g = 0; s0.live; ++g) {
            MV_WTL(0);
            if (!URES) {
                const WeightReq wr = weight_req(s1, (g + 1) & 1);
#pragma unroll
                for (int m = 0; m < NUW; ++m) weight_piece(wr, m);
            }
            const SliceReq sr = slice_req(s3, (g + 3) & 3);
#pragma unroll
            for (int n = 0; n < NIW; ++n) slice_piece(sr, n, dbl[n], dpl[n]);
            MV_WTL(1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
            MV_WTL(2);
            __builtin_amdgcn_s_barrier();
            MV_WTL(3);
            s0 = s1;
            s1 = s2;
            s2 = s3;
            s3 = next_step(s3);
        }
        return;
    }
    f32x4v acc[16][NT];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[q][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    read_block(0);
    transform_block();

    for (int g = 0; s0.live; ++g) {
        const bool last = is_last(s0);
        MV_WTL(0);
        {
            // the tile's output offsets and skip values: only needed behind its last step, requested ahead of the DMA
            // (older in the memory queue than anything this step waits for)
            unsigned ooff[2][2];
            f32x4v skv[2][2][NT];
            if (last) {
                const TilePos& here = s0.pos;
                const unsigned oorigin = (unsigned)((((here.b * a.Do + here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool ok = here.ty0 + 2 * wave + i < a.Ho && here.tx0 + 2 * lm + j < a.Wo;
                        ooff[i][j] = ok ? obase + oorigin + i * orow + j * opix : 0x80000000u;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            skv[i][j][nt] = (SKIP && AHEAD) ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooff[i][j] + nt * 64, 0, 0))
                                                            : (f32x4v){0.f, 0.f, 0.f, 0.f};
                    }
            }
            // This step's requests (weights of step g+1, then the slice of step g+3) and the 16 reads of the next step's
            // pixels are spread over the four MFMA groups: their issue stalls fall between this wave's MFMAs, where the
            // other wave of the SIMD has MFMAs to run, instead of at a place where both wait.
            const WeightReq wr = weight_req(s1, (g + 1) & 1);
            const SliceReq sr = slice_req(s3, (g + 3) & 3);
            const f32x4v* const nextp = ring + ((g + 1) & 3) * SLICE + abase;
            __builtin_amdgcn_sched_barrier(0);
            {
                const f32x4v* const us = uring + (URES ? s0.kz * NCH + s0.c : g & 1) * UST + ntw * 64 + lane;
                f32x4v ub[2][4][NT];
                auto load_u = [&](int grp, f32x4v (&dst)[4][NT]) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) dst[qq][nt] = us[((grp * 4 + qq) * NTW + nt) * 64];
                };
                constexpr int NUL = URES ? 0 : NUW;                 // weight pieces per step
                constexpr int NP = SPLITN ? NUL + NIW : 0;          // DMA pieces of a compute wave per step
                load_u(0, ub[0]);
#pragma unroll
                for (int grp = 0; grp < 4; ++grp) {
                    if (grp < 3) load_u(grp + 1, ub[(grp + 1) & 1]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[grp * 4 + qq][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[grp & 1][qq][nt][j], elem(V[grp][qq], j),
                                                                                             acc[grp * 4 + qq][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = grp * 2; k < grp * 2 + 2 && k < NP; ++k) {
                        if (k < NUL) {
                            weight_piece(wr, k);
                        } else {
                            slice_piece(sr, k - NUL, dbase[k - NUL], dpos[k - NUL]);
                        }
                    }
                    if (AHEAD) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {           // row grp of the next step's 4x4 block
                            const f32x4v v = nextp[grp * RS + (x & 1) * (G::PWH * 2) + (x >> 1) * 2];
                            d[grp][x] = {{v[0], v[1]}, {v[2], v[3]}};
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            MV_WTL(1);
            __builtin_amdgcn_sched_barrier(0);
            // (the packed additions are inline assembly: MFMA results reach them only behind these wait states, and the
            //  transform below overwrites MFMA source registers)
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (last) {
                if (SKIP && !AHEAD) {
                    // (two N tiles per wave: no registers to hold the skip values through the step -- requested here, the
                    //  output transform below covers most of their latency)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                skv[i][j][nt] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooff[i][j] + nt * 64, 0, 0));
                }
                // Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; fused epilogue on the 2x2 pixels x 4 channels of this lane
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    Q4 m[16], t[2][4];
#pragma unroll
                    for (int q = 0; q < 16; ++q) m[q] = {{acc[q][nt][0], acc[q][nt][1]}, {acc[q][nt][2], acc[q][nt][3]}};
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        t[0][x] = m[0 + x] + m[4 + x] + m[8 + x];
                        t[1][x] = m[4 + x] - (m[8 + x] + m[12 + x]);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        Q4 y[2];
                        y[0] = t[i][0] + t[i][1] + t[i][2];
                        y[1] = t[i][1] - (t[i][2] + t[i][3]);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            f32x4v v = {y[j].lo[0], y[j].lo[1], y[j].hi[0], y[j].hi[1]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = fmaxf(fmaf(v[e], scv[nt][e], shv[nt][e]), floor_v);
                                if (SKIP) v[e] += skv[i][j][nt][e];
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, ooff[i][j] + nt * 64, 0, MV_STORE_AUX);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[q][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!AHEAD) read_block((g + 1) & 3);
            transform_block();                // d (step g+1) -> V
            MV_WTL(2);
            __builtin_amdgcn_sched_barrier(0);
            // slice g+2 and weights g+1 have landed once only this step's slice request (and the stores behind it) are
            // outstanding
            if (SPLITN) {
                if (last) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW + 4 * NT) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
                }
            }
        }
        // everyone is done with slice g and weight slot g & 1
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        MV_WTL(3);
        s0 = s1;
        s1 = s2;
        s2 = s3;
        s3 = next_step(s3);
    }
}

template <int NCH, int KD, bool SKIP, int MODE>
int launch_wino_ring(const ConvArgs& a, hipStream_t s) {
    using G = RingGeom;
    constexpr int NT = MODE == 0 ? 1 : 2;                  // N tiles per workgroup
    constexpr int USLOTS = KD * NCH * NT <= 4 ? KD * NCH : 2;      // resident weights, else a ring of two steps
    const size_t lds = (size_t)(4 * G::SLICE + USLOTS * 16 * NT * 64 + 64) * 16;
    if (lds > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv_wino_ring_kernel<NCH, KD, SKIP, MODE>;
    static unsigned long attr_done = 0;
    if (!allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    PersArgs p;
    if (!fill_pers_args(a, G::TY, p)) return MVSTER_ERR_UNSUPPORTED;
    if ((long)KD * 16 * NCH * a.ntile_total * 1024 >= (1L << 31)) return MVSTER_ERR_UNSUPPORTED;
    const long ntiles = p.ntiles;
    const int ny = a.ntile_total / NT;
    long gmax = (long)ncu / ny;                            // one workgroup per CU (LDS)
    if (gmax < 1) gmax = 1;
    const long rounds = (ntiles + gmax - 1) / gmax;       // equal shares
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv_wino_ring_kernel<%d, %d, %s, %d>", NCH, KD, SKIP ? "true" : "false", MODE);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, ny, 1), dim3(512), lds, s, a, p);
    return mv_check_launch();
}

// ------------------------------------------------------------------------------------------------------------------
// Pair form of the 3x3x3 layers (round 6, review item 4: "transform each input slice once").  The ring kernel above walks
// (output slice z, depth tap kz, chunk): input slice z + kz - 1 is fetched and transformed once per (z, kz) -- three times.
// Walking the input slices once needs three live accumulator sets (192 registers beside V's 64 and the weight read-ahead:
// does not fit at two waves per SIMD, scripts/probes/wino_once_regs.hip).  TWO sets fit -- the register budget of MODE 2,
// with the second output SLICE where MODE 2 holds the second N tile: a tile is the output slices (2 zp, 2 zp + 1) of an
// 8 x 32 window and its steps walk the four input slices 2 zp - 1 .. 2 zp + 2 (window index kz = 0..3) x chunks; the slice of
// window index kz feeds set 0 (output 2 zp) through depth tap kz and set 1 (output 2 zp + 1) through tap kz - 1.  Four fetches
// and transforms per two output slices instead of six (D = 4 with its padding: six instead of ten), and 128 MFMAs behind the
// inner transforms instead of 64.  Per output element the (depth tap, chunk) summation order is the ring kernel's: results are
// bit-identical to it.  Frame = MODE 0 / 2 of the ring kernel: waves 0-3 compute (one N tile, blockIdx.y picks it), waves 4-7
// issue the DMA; 16-channel inputs keep their three weight blocks resident, 32-channel inputs stream two blocks per step.
// ------------------------------------------------------------------------------------------------------------------
template <int NCH, bool SKIP>
__global__ void __launch_bounds__(512) conv_wino_pair_kernel(ConvArgs a, PersArgs p) {
    using G = RingGeom;
    constexpr int KD = 3, NS = 2;
    constexpr int TY = G::TY, PLANE = G::PLANE, NBLK = G::NBLK, RS = G::ROWSLOTS, SLICE = G::SLICE;
    constexpr int NLW = 4;                                  // loading waves
    constexpr int NIW = (NBLK + NLW / 2 - 1) / (NLW / 2);  // slice DMA instructions per loading wave and step
    constexpr int CIN = NCH * 16;
    constexpr int UBLK = 16 * 64;                           // float4 of one weight block: (depth tap, chunk) of one N tile
    constexpr bool URES = KD * NCH <= 3;                    // every block of the workgroup's N tile stays in LDS
    constexpr int UST = URES ? UBLK : NS * UBLK;            // float4 per weight slot
    constexpr int USLOTS = URES ? KD * NCH : 2;
    constexpr int NUW = NS * 16 / NLW;                      // weight DMA instructions per loading wave and step (streamed form)
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const ring = reinterpret_cast<f32x4v*>(lds_raw);              // 4 patch slices
    f32x4v* const uring = ring + 4 * SLICE;
    f32x4v* const scratch = uring + USLOTS * UST;                         // 64 float4: surplus DMA slots

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave8 >> 2, wave = wave8 & 3;
    const int lm = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.y;
    const long wstep = (long)a.ntile_total * 256;          // floats per K step (16 channels) of the packed weights
    const int dop = a.Do >> 1;                              // output slice pairs
    auto in_rsrc = [&](int k) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in) - 256 * k, (short)0, (int)(a.in_bytes + 1024u * k), 0x00020000);
    };
    auto w_rsrc = [&](int k) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk) - 256 * k, (short)0,
                                                 (int)(KD * 16 * NCH * wstep * 4 + 1024 * k), 0x00020000);
    };
    const int lpl = wave / (NLW / 2), lblk0 = (wave % (NLW / 2)) * NIW;   // a loading wave's plane and first block
    auto decode_piece = [&](int n, int ln, unsigned& db, int& dp) {
        const int blk = lblk0 + n;
        const int s = blk * 64 + ln;
        const int q1 = s & 1;
        int t = s >> 1;
        const int xh = t % G::PWH;
        t /= G::PWH;
        const int py = t >> 1, px = 2 * xh + (t & 1);
        const bool valid = blk < NBLK && py < G::PH;
        dp = px | (py << 8);
        db = valid ? (unsigned)((py * a.Wi + px) * (CIN * 4) + (lpl * 8 + q1 * 4) * 4) : 0x80000000u;
    };
    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        auto div = [&](unsigned n, int k) -> unsigned { return ((__umulhi(n, p.mul[k]) >> p.shr[k]) & ~p.one[k]) | (n & p.one[k]); };
        unsigned q = div(tile, 0);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = div(q, 1);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = div(q2, 2);                     // (the host formed this divisor from Do / 2)
        t.zo = (int)(q2 - q3 * (unsigned)dop);              // pair index
        t.b = (int)q3;
        return t;
    };
    const unsigned nwg = gridDim.x;
    // a step = (window slice kz = 0..3, chunk c); window slices outside the volume have no step
    auto first_step = [&](unsigned tile) -> RingStep {
        RingStep st;
        st.tile = tile;
        st.live = tile < p.ntiles;
        st.pos = decode_tile(st.live ? tile : 0u);
        st.kz = max(0, 1 - 2 * st.pos.zo);
        st.kz_hi = min(4, a.Di - 2 * st.pos.zo + 1);
        st.c = 0;
        return st;
    };
    auto is_last = [&](const RingStep& st) -> bool { return st.c == NCH - 1 && st.kz == st.kz_hi - 1; };
    auto next_step = [&](const RingStep& st) -> RingStep {
        if (!st.live) return st;
        if (is_last(st)) return first_step(st.tile + nwg);
        RingStep n = st;
        if (++n.c == NCH) {
            n.c = 0;
            ++n.kz;
        }
        return n;
    };
    struct SliceReq {
        unsigned origin, wi;
        int iy0, ix0;
        bool interior;
        f32x4v* dst0;
    };
    auto slice_req = [&](const RingStep& st, int slot) -> SliceReq {
        SliceReq r;
        const int iz = 2 * st.pos.zo + st.kz - 1;
        r.iy0 = st.pos.ty0 - a.ph[0];
        r.ix0 = st.pos.tx0 - a.pw[0];
        r.origin = (unsigned)(((((st.pos.b * a.Di + iz) * a.Hi + r.iy0) * a.Wi + r.ix0) * CIN + st.c * 16) * 4);
        r.wi = st.live ? (unsigned)a.Wi : 0u;
        r.interior = st.live && r.iy0 >= 0 && r.ix0 >= 0 && r.iy0 + G::PH <= a.Hi && r.ix0 + 2 * G::PWH <= a.Wi;
        r.dst0 = ring + slot * SLICE;
        return r;
    };
    auto slice_piece = [&](const SliceReq& r, int n, unsigned db, int dp) {
        constexpr int IMM[4] = {0, 1024, 2048, 3072};
        const bool real = lblk0 + n < NBLK;
        f32x4v* const dst = (NIW * (NLW / 2) == NBLK || n + 1 < NIW || real) ? r.dst0 + lpl * PLANE + (lblk0 + (n & ~3)) * 64
                                                                             : scratch - (n & 3) * 64;
        const __amdgpu_buffer_rsrc_t rs = in_rsrc(n & 3);
        if (r.interior) {
            const unsigned so = r.origin;
            switch (n & 3) {
                case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[0], 0); break;
                case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[1], 0); break;
                case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[2], 0); break;
                default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, db, so, IMM[3], 0); break;
            }
        } else {
            const int ix = r.ix0 + (dp & 255), iy = r.iy0 + (dp >> 8);
            const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < r.wi;
            const unsigned off = ok ? db + r.origin : 0x80000000u;
            switch (n & 3) {
                case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[0], 0); break;
                case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[1], 0); break;
                case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[2], 0); break;
                default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, IMM[3], 0); break;
            }
        }
    };
    // one kilobyte of a weight block: transform point q of (depth tap kz, chunk c), to `dst` (+ the immediate of piece m & 3)
    const unsigned wlane = lane * 16;
    auto weight_piece = [&](int kz, int c, int q, f32x4v* dst, int m) {
        const unsigned so = (unsigned)((((long)(kz * 16 + q) * NCH + c) * wstep + (long)nt0 * 256) * 4);
        const __amdgpu_buffer_rsrc_t rs = w_rsrc(m & 3);
        switch (m & 3) {
            case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 0, 0); break;
            case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 1024, 0); break;
            case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 2048, 0); break;
            default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, wlane, so, 3072, 0); break;
        }
    };
    // streamed weights of one step: slot = [set][point][lane]; this wave's NUW kilobytes are consecutive and belong to ONE set
    // (set s takes depth tap kz - s; a set without a tap in this step fetches nothing)
    auto weight_step = [&](const RingStep& st, int slot) {
        const int set = (wave * NUW) >> 4, q0 = (wave * NUW) & 15;
        const int kz = st.kz - set;
        if (!st.live || kz < 0 || kz >= KD) return;
        f32x4v* const dst0 = uring + slot * UST + wave * NUW * 64;
#pragma unroll
        for (int m = 0; m < NUW; ++m) weight_piece(kz, st.c, q0 + m, dst0 + (m & ~3) * 64, m);
    };

    RingStep s0 = first_step(xcd_remap(blockIdx.x, nwg));
    RingStep s1 = next_step(s0), s2 = next_step(s1), s3 = next_step(s2);
    if (half == 1) {
        // ---- loading waves: the first three slices and the first weights, then one request set per step
        unsigned dbl[NIW];
        int dpl[NIW];
#pragma unroll
        for (int n = 0; n < NIW; ++n) decode_piece(n, lane, dbl[n], dpl[n]);
        const RingStep* const first[3] = {&s0, &s1, &s2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const SliceReq r = slice_req(*first[k], k);
#pragma unroll
            for (int n = 0; n < NIW; ++n) slice_piece(r, n, dbl[n], dpl[n]);
        }
        if (URES) {
#pragma unroll
            for (int kz = 0; kz < KD; ++kz)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int m = 0; m < 4; ++m)         // 16 points over four waves
                        weight_piece(kz, c, wave * 4 + m, uring + (kz * NCH + c) * UBLK + wave * 4 * 64, m);
        } else {
            weight_step(s0, 0);
        }
        __syncthreads();        // (waits for everything requested so far)
        for (int g = 0; s0.live; ++g) {
            if (!URES) weight_step(s1, (g + 1) & 1);
            const SliceReq sr = slice_req(s3, (g + 3) & 3);
#pragma unroll
            for (int n = 0; n < NIW; ++n) slice_piece(sr, n, dbl[n], dpl[n]);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
            __builtin_amdgcn_s_barrier();
            s0 = s1;
            s1 = s2;
            s2 = s3;
            s3 = next_step(s3);
        }
        return;
    }
    __syncthreads();

    // ---- compute waves ------------------------------------------------------------------------------------------------
    const int n0 = nt0 * 16 + lq * 4;
    const f32x4v scv = *reinterpret_cast<const f32x4v*>(a.scale + n0);
    const f32x4v shv = *reinterpret_cast<const f32x4v*>(a.shift + n0);
    const int abase = 2 * wave * RS + 2 * lm + (lq >> 1) * PLANE + (lq & 1);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)p.out_bytes : 0, 0x00020000);
    const unsigned obase = (unsigned)((2 * wave * a.Wo + 2 * lm) * a.cout + nt0 * 16 + lq * 4) * 4u;
    const unsigned opix = (unsigned)a.cout * 4u, orow = (unsigned)a.Wo * opix;
    const unsigned oslice = (unsigned)a.Ho * orow;
    const float floor_v = a.relu ? 0.0f : -__builtin_inff();

    Q4 d[4][4], V[4][4];
    auto read_block = [&](int slot) {
        const f32x4v* patch = ring + slot * SLICE + abase;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const f32x4v v = patch[r * RS + (x & 1) * (G::PWH * 2) + (x >> 1) * 2];
                d[r][x] = {{v[0], v[1]}, {v[2], v[3]}};
            }
    };
    auto transform_block = [&]() {       // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const Q4 t0 = d[0][x] - d[2][x], t1 = d[1][x] + d[2][x], t2 = d[2][x] - d[1][x], t3 = d[1][x] - d[3][x];
            d[0][x] = t0;
            d[1][x] = t1;
            d[2][x] = t2;
            d[3][x] = t3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            V[r][0] = d[r][0] - d[r][2];
            V[r][1] = d[r][1] + d[r][2];
            V[r][2] = d[r][2] - d[r][1];
            V[r][3] = d[r][1] - d[r][3];
        }
    };
    f32x4v acc[16][NS];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int st = 0; st < NS; ++st) acc[q][st] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    read_block(0);
    transform_block();

    for (int g = 0; s0.live; ++g) {
        const bool last = is_last(s0);
        // weight blocks of this step's sets: resident [tap][chunk] or this step's slot [set]
        const f32x4v* const u0 = uring + (URES ? (min(s0.kz, KD - 1) * NCH + s0.c) * UBLK : (g & 1) * UST) + lane;
        const f32x4v* const u1 = uring + (URES ? (max(s0.kz - 1, 0) * NCH + s0.c) * UBLK : (g & 1) * UST + UBLK) + lane;
        __builtin_amdgcn_sched_barrier(0);
        // One set after the other, each behind ONE wave-uniform branch: a set without a depth tap in this step (the window's
        // first slice has none for set 1, its last none for set 0) is skipped.  Weights are read one group of four transform
        // points ahead.  (Three copies of a two-set phase, or branches around groups of MFMAs, made the allocator spill.)
        auto set_phase = [&](auto st_c, const f32x4v* us) {
            constexpr int ST = decltype(st_c)::value;
            f32x4v ub[2][4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) ub[0][qq] = us[qq * 64];
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                if (grp < 3) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) ub[(grp + 1) & 1][qq] = us[((grp + 1) * 4 + qq) * 64];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        acc[grp * 4 + qq][ST] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[grp & 1][qq][j], elem(V[grp][qq], j),
                                                                                     acc[grp * 4 + qq][ST], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (s0.kz < KD) set_phase(std::integral_constant<int, 0>{}, u0);
        __builtin_amdgcn_sched_barrier(0);
        if (s0.kz > 0) set_phase(std::integral_constant<int, 1>{}, u1);
        __builtin_amdgcn_sched_barrier(0);
        // (the packed additions are inline assembly: MFMA results reach them only behind these wait states, and the
        //  transform below overwrites MFMA source registers)
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (last) {
            const TilePos& here = s0.pos;
            const unsigned oorigin = (unsigned)((((here.b * a.Do + 2 * here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
            unsigned ooff[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool ok = here.ty0 + 2 * wave + i < a.Ho && here.tx0 + 2 * lm + j < a.Wo;
                    ooff[i][j] = ok ? obase + oorigin + i * orow + j * opix : 0x80000000u;
                }
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                f32x4v skv[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const unsigned vo = ooff[i][j] + st * oslice;
                        skv[i][j] = SKIP ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, vo, 0, 0))
                                         : (f32x4v){0.f, 0.f, 0.f, 0.f};
                    }
                // Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; fused epilogue on the 2x2 pixels x 4 channels of this lane
                Q4 m[16], t[2][4];
#pragma unroll
                for (int q = 0; q < 16; ++q) m[q] = {{acc[q][st][0], acc[q][st][1]}, {acc[q][st][2], acc[q][st][3]}};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    t[0][x] = m[0 + x] + m[4 + x] + m[8 + x];
                    t[1][x] = m[4 + x] - (m[8 + x] + m[12 + x]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    Q4 y[2];
                    y[0] = t[i][0] + t[i][1] + t[i][2];
                    y[1] = t[i][1] - (t[i][2] + t[i][3]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        f32x4v v = {y[j].lo[0], y[j].lo[1], y[j].hi[0], y[j].hi[1]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = fmaxf(fmaf(v[e], scv[e], shv[e]), floor_v);
                            if (SKIP) v[e] += skv[i][j][e];
                        }
                        // (the slice offset goes into the VECTOR offset: with an SGPR soffset hipcc 7.2 puts the next output's
                        //  v_fma right behind the store -- LLVM's hazard recogniser only separates a > 64-bit store from a VALU
                        //  write of its data registers when soffset is NOT a register -- and gfx950 then stores the new value in
                        //  the last lanes of the second dword: one wrong element in 128, found by the bit-equality check)
                        const unsigned vo = ooff[i][j] + st * oslice;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, vo, 0, MV_STORE_AUX);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int st = 0; st < NS; ++st) acc[q][st] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
        read_block((g + 1) & 3);
        transform_block();                // d (step g+1) -> V
        __builtin_amdgcn_sched_barrier(0);
        // everyone is done with slice g and weight slot g & 1
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        s0 = s1;
        s1 = s2;
        s2 = s3;
        s3 = next_step(s3);
    }
}

template <int NCH, bool SKIP>
int launch_wino_pair(const ConvArgs& a, hipStream_t s) {
    using G = RingGeom;
    constexpr int KD = 3;
    constexpr bool URES = KD * NCH <= 3;
    constexpr int USLOTS = URES ? KD * NCH : 2, UST = (URES ? 1 : 2) * 16 * 64;
    const size_t lds = (size_t)(4 * G::SLICE + USLOTS * UST + 64) * 16;
    if (lds > 160 * 1024 || (a.Do & 1) || a.Do != a.Di) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv_wino_pair_kernel<NCH, SKIP>;
    static unsigned long attr_done = 0;
    if (!allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    ConvArgs pairs = a;                                    // tiles = (b, slice PAIR, 8 x 32 window)
    pairs.Do = a.Do / 2;
    PersArgs p;
    if (!fill_pers_args(pairs, G::TY, p)) return MVSTER_ERR_UNSUPPORTED;
    if ((long)KD * 16 * NCH * a.ntile_total * 1024 >= (1L << 31)) return MVSTER_ERR_UNSUPPORTED;
    const long ntiles = p.ntiles;
    const int ny = a.ntile_total;
    long gmax = (long)ncu / ny;                            // one workgroup per CU (LDS)
    if (gmax < 1) gmax = 1;
    const long rounds = (ntiles + gmax - 1) / gmax;       // equal shares
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv_wino_pair_kernel<%d, %s>", NCH, SKIP ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, ny, 1), dim3(512), lds, s, a, p);
    return mv_check_launch();
}

// G g G^T for one (cout, cin) pair, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]; one thread per packed element
struct PackWinoDesc {          // 88 bytes; the batched form reads a device table of these
    const float* w;
    float* wpk;
    long s_n, s_c, s_z, s_y, s_x;
    int cout, cin_raw, cin, kd, flip, ntile;
    int first_block, pad_;     // first_block: prefix sum of ceil(total / 256) over the records before this one
};

__device__ __forceinline__ void pack_wino_element(const PackWinoDesc& d, long idx) {
    const int cin = d.cin, kd = d.kd, ntile = d.ntile, flip = d.flip;
    const long total = (long)kd * 16 * cin * ntile * 16;
    if (idx >= total) return;
    // packed fragment order [K step][N tile][lane][j]: K = (kz * 16 + point) * cin + ci, lane = (ci % 16 / 4) * 16 + (n % 16), j = ci % 4
    const int j = (int)(idx & 3), lanei = (int)((idx >> 2) & 63);
    const long rest = idx >> 8;
    const int t = (int)(rest % ntile);
    const int kstep = (int)(rest / ntile);
    const int k = kstep * 16 + (lanei >> 4) * 4 + j;
    const int n = t * 16 + (lanei & 15);
    const int qz = k / cin, ci = k - qz * cin;
    const int kz = qz >> 4, q = qz & 15;
    float u = 0.f;
    if (n < d.cout && ci < d.cin_raw) {
        const int xi = q >> 2, nu = q & 3;
        const float Gm[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        const float* g = d.w + n * d.s_n + ci * d.s_c + (flip ? kd - 1 - kz : kz) * d.s_z;
        float rowv[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float acc = 0.f;
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                const int yy = flip ? 2 - y : y, xx = flip ? 2 - x : x;
                acc = fmaf(Gm[xi][y], g[yy * d.s_y + xx * d.s_x], acc);
            }
            rowv[x] = acc;
        }
        u = fmaf(Gm[nu][0], rowv[0], fmaf(Gm[nu][1], rowv[1], Gm[nu][2] * rowv[2]));
    }
    d.wpk[idx] = u;
}

__global__ void pack_wino_kernel(PackWinoDesc d) { pack_wino_element(d, (long)blockIdx.x * blockDim.x + threadIdx.x); }

// every record of a device table in ONE launch (the training step re-transforms the weights of all its Winograd layers,
// forward and input-gradient forms, after each optimizer update: 44 launches of ~5 us as separate calls)
__global__ void __launch_bounds__(256) pack_wino_batch_kernel(const PackWinoDesc* __restrict__ descs, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackWinoDesc d = descs[lo];
    pack_wino_element(d, (long)((int)blockIdx.x - d.first_block) * 256 + threadIdx.x);
}

}  // namespace

// variant 8: 3x3 stride-1 pad-1 single-slice convolutions, cin in {16, 32}, weights resident per workgroup;
// variant 9 (ring = true): (1|3)x3x3 stride 1, pad (kd/2, 1, 1), cin in {16, 32, 64}, patch slices and weights streamed.
// cout % 16 == 0, optional same-shape skip.
int dispatch_wino(const ConvArgs& a, int nt, int wpc, bool ring, hipStream_t s) {
    if (a.nclass != 1 || a.osd != 1 || a.osh != 1 || a.osw != 1 || a.skip_mode > 1 || a.prob_w || a.cout % 16 != 0 ||
        a.sh != 1 || a.sw != 1 || a.sd != 1 || a.kh[0] != 3 || a.kw[0] != 3 || a.ph[0] != 1 || a.pw[0] != 1 ||
        nt < 1 || a.ntile_total % nt != 0 || a.cin % 16 != 0)
        return MVSTER_ERR_UNSUPPORTED;
    const int nch = a.cin / 16, kd = a.kd[0];
    if (!(kd == 1 && a.pd[0] == 0) && !(kd == 3 && a.pd[0] == 1)) return MVSTER_ERR_UNSUPPORTED;
    if (ring) {
        // nt = 1: waves 4-7 only issue the DMA (mode 0); nt = 2: the two halves of the workgroup compute one N tile each
        // (mode 1), or with wpc = 1 the compute waves hold both and waves 4-7 load (mode 2)
        if (nt == 1 && wpc == 3) {                         // pair form: two output slices per tile (kd == 3, 16 / 32 channels in)
            if (kd != 3) return MVSTER_ERR_UNSUPPORTED;
            if (nch == 1) return a.skip_mode == 1 ? launch_wino_pair<1, true>(a, s) : launch_wino_pair<1, false>(a, s);
            if (nch == 2) return a.skip_mode == 1 ? launch_wino_pair<2, true>(a, s) : launch_wino_pair<2, false>(a, s);
            return MVSTER_ERR_UNSUPPORTED;
        }
        const int mode = nt == 1 ? 0 : (wpc == 1 ? 2 : 1);
#define MV_R(MODE_, NCH_, KD_)                                                                      \
    if (mode == MODE_ && nch == NCH_ && kd == KD_)                                                  \
        return a.skip_mode == 1 ? launch_wino_ring<NCH_, KD_, true, MODE_>(a, s) : launch_wino_ring<NCH_, KD_, false, MODE_>(a, s);
        MV_R(0, 1, 3) MV_R(0, 2, 3) MV_R(1, 2, 3) MV_R(2, 2, 3) MV_R(0, 4, 3) MV_R(1, 4, 3) MV_R(2, 4, 3)
        MV_R(0, 4, 1) MV_R(1, 4, 1) MV_R(2, 4, 1) MV_R(0, 2, 1) MV_R(1, 2, 1) MV_R(2, 2, 1) MV_R(0, 1, 1) MV_R(2, 1, 1)
#undef MV_R
        return MVSTER_ERR_UNSUPPORTED;
    }
    if (kd != 1) return MVSTER_ERR_UNSUPPORTED;
#define MV_W(NT_, NCH_, WREG_)                                                                      \
    if (nt == NT_ && nch == NCH_)                                                                   \
        return a.skip_mode == 1 ? launch_wino<NT_, NCH_, WREG_, true>(a, wpc, s) : launch_wino<NT_, NCH_, WREG_, false>(a, wpc, s);
    MV_W(1, 1, true)       // 16 -> 16
    MV_W(2, 1, true)       // 16 -> 32
    MV_W(2, 2, false)      // 32 -> 32 (U: 64 KB of LDS)
    MV_W(1, 2, false)
#undef MV_W
    return MVSTER_ERR_UNSUPPORTED;
}

}  // namespace mvconv

#ifdef MVSTER_TIMELINE
extern "C" int mvster_debug_wino_timeline(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(mvconv::g_wtl), &buf, sizeof(buf)) == hipSuccess ? MVSTER_OK : MVSTER_ERR_LAUNCH;
}
#endif

// Transformed weights for variants 8 / 9: w [cout, cin, kd, 3, 3] (element strides given; `flip` mirrors the taps, for the
// input-gradient form) -> wpk [kd * 16 * cin_pad / 16][ceil(cout / 16)][64][4] floats.
extern "C" int mvster_pack_wino_weights(const float* w, float* wpk, int cout, int cin_raw, int cin_pad, int kd, long s_n, long s_c,
                                        long s_z, long s_y, long s_x, int flip, void* stream) {
    if (!w || !wpk) return MVSTER_ERR_NULL;
    if (cout < 1 || cin_raw < 1 || cin_pad < cin_raw || cin_pad % 16 != 0 || (kd != 1 && kd != 3)) return MVSTER_ERR_SHAPE;
    const int ntile = (cout + 15) / 16;
    const long total = (long)kd * 16 * cin_pad * ntile * 16;
    mvconv::PackWinoDesc d{w, wpk, s_n, s_c, s_z, s_y, s_x, cout, cin_raw, cin_pad, kd, flip, ntile, 0, 0};
    hipLaunchKernelGGL(mvconv::pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d);
    return mv_check_launch();
}

// The same for every record of a DEVICE table in one launch: 88-byte records {const float* w; float* wpk; long s_n, s_c,
// s_z, s_y, s_x; int cout, cin_raw, cin_pad, kd, flip, ntile, first_block, pad} with ntile = ceil(cout / 16) and
// first_block = prefix sum of ceil(kd * 16 * cin_pad * ntile * 16 / 256); total_blocks = their sum.
extern "C" int mvster_pack_wino_batch(const void* descs, int ndesc, int total_blocks, void* stream) {
    if (!descs) return MVSTER_ERR_NULL;
    if (ndesc < 1 || total_blocks < 1) return MVSTER_ERR_SHAPE;
    static_assert(sizeof(mvconv::PackWinoDesc) == 88, "PackWinoDesc layout");
    hipLaunchKernelGGL(mvconv::pack_wino_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const mvconv::PackWinoDesc*)descs, ndesc);
    return mv_check_launch();
}
