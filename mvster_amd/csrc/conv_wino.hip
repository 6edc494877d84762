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

// Packed fp32 additions: one instruction per TWO lanes-elements (the transforms are additions only; hipcc emits v_pk_add_f32
// for a + b but four v_sub_f32 for a float4 subtraction -- the negation is an operand modifier of the packed form).
// The hazard recogniser does not look inside inline assembly: results of MFMAs reach these only behind the explicit
// s_nop block after the MFMA phase (see the main loop).
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
        const int py = pix / PW, px = pix - py * PW;
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
    // float4 index of this lane's 4x4 block origin: patch row 2*wave, column 2*lm, its channel quad
    const int abase = 2 * wave * RS + 4 * lm + (lq >> 1) * PLANE + (lq & 1);
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
                    const f32x4v v = patch[c * 2 * PLANE + r * RS + x * 2];
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
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, ooff[i][j] + nt * 64, 0, 0);
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
    static bool attr_set = false;
    if (!attr_set) {
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return MVSTER_ERR_LAUNCH;
        attr_set = true;
    }
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

// G g G^T for one (cout, cin) pair, G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]; one thread per packed element
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ wpk, int cout, int cin_raw, int cin, long s_n,
                                 long s_c, long s_y, long s_x, int flip, int ntile) {
    const long total = (long)16 * cin * ntile * 16;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // packed fragment order [K step][N tile][lane][j]: K = point * cin + ci, lane = (ci % 16 / 4) * 16 + (n % 16), j = ci % 4
    const int j = (int)(idx & 3), lanei = (int)((idx >> 2) & 63);
    const long rest = idx >> 8;
    const int t = (int)(rest % ntile);
    const int kstep = (int)(rest / ntile);
    const int k = kstep * 16 + (lanei >> 4) * 4 + j;
    const int n = t * 16 + (lanei & 15);
    const int q = k / cin, ci = k - q * cin;
    float u = 0.f;
    if (n < cout && ci < cin_raw) {
        const int xi = q >> 2, nu = q & 3;
        const float Gm[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
        const float* g = w + n * s_n + ci * s_c;
        float rowv[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float acc = 0.f;
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                const int yy = flip ? 2 - y : y, xx = flip ? 2 - x : x;
                acc = fmaf(Gm[xi][y], g[yy * s_y + xx * s_x], acc);
            }
            rowv[x] = acc;
        }
        u = fmaf(Gm[nu][0], rowv[0], fmaf(Gm[nu][1], rowv[1], Gm[nu][2] * rowv[2]));
    }
    wpk[idx] = u;
}

}  // namespace

// 3x3 stride-1 pad-1 single-slice convolutions, cin in {16, 32}, cout % 16 == 0, optional same-shape skip (variant 8)
int dispatch_wino(const ConvArgs& a, int nt, int wpc, hipStream_t s) {
    if (a.nclass != 1 || a.osd != 1 || a.osh != 1 || a.osw != 1 || a.skip_mode > 1 || a.prob_w || a.cout % 16 != 0 ||
        a.sh != 1 || a.sw != 1 || a.sd != 1 || a.kd[0] != 1 || a.kh[0] != 3 || a.kw[0] != 3 || a.ph[0] != 1 || a.pw[0] != 1 ||
        a.pd[0] != 0 || nt < 1 || a.ntile_total % nt != 0 || a.cin % 16 != 0)
        return MVSTER_ERR_UNSUPPORTED;
    const int nch = a.cin / 16;
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

// Transformed weights for variant 8: w [cout, cin, 3, 3] (element strides given; `flip` mirrors the taps, for the
// input-gradient form) -> wpk [16 * cin_pad / 16][ceil(cout / 16)][64][4] floats.
extern "C" int mvster_pack_wino_weights(const float* w, float* wpk, int cout, int cin_raw, int cin_pad, long s_n, long s_c, long s_y,
                                        long s_x, int flip, void* stream) {
    if (!w || !wpk) return MVSTER_ERR_NULL;
    if (cout < 1 || cin_raw < 1 || cin_pad < cin_raw || cin_pad % 16 != 0) return MVSTER_ERR_SHAPE;
    const int ntile = (cout + 15) / 16;
    const long total = (long)16 * cin_pad * ntile * 16;
    hipLaunchKernelGGL(mvconv::pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, wpk, cout,
                       cin_raw, cin_pad, s_n, s_c, s_y, s_x, flip, ntile);
    return mv_check_launch();
}
