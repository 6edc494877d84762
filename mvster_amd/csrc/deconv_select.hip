// reg2d's last layer + the depth selection on the fp32 matrix cores (round 4): ConvTranspose (1,3,3) stride (1,2,2) 16 -> 8 +
// BatchNorm + ReLU + skip, the 1x1x1 `prob` head, softmax over depth, first-max argmax, gather, confidence, inverse bounds
// (models/mvs4net_utils.py:897-900, :1068-1088) in one persistent launch.
//
// Why: deconv_select_kernel (conv_small.hip) walks the 16 input channels in a rolled loop of dependent gathers on the VALU:
// 17-20 us for the three coarse stages' tiny maps (latency) and 43 us at stage 4 for 75 MB (0.16 of the HBM peak over the
// four launches).  As a GEMM the layer is M = input voxels, N = 32 = (output parity class q = 2 dy + dx, co), K = 64 =
// (2 x 2 input neighbourhood block (di, dj)) x 16 channels; class q takes block (di, dj) through kernel tap
// (dy + 1 - 2 di, dx + 1 - 2 dj) when that tap exists, else a structural zero.  K is walked channel-major, block-minor:
// a K step of 4 = the four blocks of ONE channel, lane group lq = block -- the summation order of the VALU kernel's FMA
// chain (channel by channel, blocks (0,0), (0,1), (1,0), (1,1)), so the logits keep their bits.  32 MFMAs per 16 voxels
// (64 output pixels): 9 us of pipe time at stage 4 against 12 us of HBM time.
//
// Frame (as conv_narrow.hip): 4 compute + 4 loading waves, persistent over tiles of RI input rows x 16 input voxels x ALL D
// hypothesis slices of a batch item; the loading waves stream the (RI + 1) x 17-voxel input patches and the 2 RI x 32-pixel
// skip tiles of the D slices through a ring of R LDS stages by LDS-DMA, one tile ahead.  Compute wave w owns slices
// d = w (mod 4): per (slice, row) one M tile of 16 voxels, two accumulators (dy = 0, 1); epilogue = scale/shift, ReLU, skip
// (from LDS), `prob` dot product (the two channel halves meet through one cross-lane exchange), logits to LDS.  After the
// tile's barrier thread t finishes output pixel t of the 2 RI x 32 tile with the shared mv::select_from_logits arithmetic
// (hypotheses fetched ahead of the MFMA phase); the logits buffer is double-buffered, so one barrier per tile.
// LDS patch layout: [slice][row][voxel][quad ^ ((voxel >> 2) & 3)] -- conflict-free 16-lane groups for the operand reads.
#include "conv_args.hpp"

namespace {

using mvconv::f32x4v;
using mvconv::lds_void;
using mvconv::u32x4v;

struct DselArgs {
    const float *in, *w, *scale, *shift, *skip, *prob_w, *prob_b, *hypo;
    float *attn, *depth, *conf, *inv_min, *inv_max, *logits_out;
    int B, Hi, Wi, relu;
    float split_itv;
    unsigned in_bytes, skip_bytes, ntiles;
    FastDiv tiles_x, tiles_y;
};

template <int D, int RI, int R>
struct DselGeom {
    static constexpr int PSL = (RI + 1) * 17 * 4;             // patch slots (float4) of one slice
    static constexpr int PSLOTS = D * PSL;
    static constexpr int NBLK = (PSLOTS + 63) / 64;
    static constexpr int SBLK = D * 2 * RI;                   // skip tile: D slices x 2 RI rows x (32 pixels x 2 quads)
    static constexpr int NI = NBLK + SBLK;
    static constexpr int NIW = (NI + 3) / 4;
    static constexpr int STAGE = NI * 64;
    static constexpr int LG = D * 2 * RI * 32;                // floats of one logits buffer
    static constexpr size_t LDS = (size_t)(R * STAGE + 64) * 16 + (size_t)2 * LG * 4;
};

template <int D, int RI, int R>
__global__ void __launch_bounds__(512) deconv_select_mfma_kernel(DselArgs a) {
    using G = DselGeom<D, RI, R>;
    constexpr int NBLK = G::NBLK, NI = G::NI, NIW = G::NIW, PSL = G::PSL;
    constexpr int SPW = D / 4;                                 // slices per compute wave
    static_assert(D % 4 == 0 && D <= mv::kSelMaxD && RI <= 4 && (R - 2) * NIW <= 63, "");
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const scratch = lds + R * G::STAGE;
    float* const lgbuf = lds_raw + (size_t)(R * G::STAGE + 64) * 4;

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const int lm = lane & 15, lq = lane >> 4;
    const int Ho = 2 * a.Hi, Wo = 2 * a.Wi;
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);

    auto decode = [&](unsigned t, int& b, int& i0, int& j0) {
        unsigned txu, tyu;
        b = (int)fdivmod(fdivmod(t, a.tiles_x, txu), a.tiles_y, tyu);
        i0 = (int)tyu * RI;
        j0 = (int)txu * 16;
    };

    if (loader) {
        const __amdgpu_buffer_rsrc_t in_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.skip ? a.skip : a.in), (short)0, a.skip ? (int)a.skip_bytes : 0, 0x00020000);
        unsigned dbase[NIW];
        int dpos[NIW];
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int i = wave + 4 * n;
            if (i < NBLK) {
                const int s = i * 64 + lane;
                const int d = s / PSL, r1 = s - d * PSL;
                const int prow = r1 / 68, r2 = r1 - prow * 68;
                const int v = r2 >> 2, quad = (r2 & 3) ^ ((v >> 2) & 3);
                dpos[n] = v | (prow << 8);
                dbase[n] = s < G::PSLOTS ? (unsigned)((((d * a.Hi + prow) * a.Wi + v) * 16 + quad * 4) * 4) : 0x80000000u;
            } else {
                const int s = (i - NBLK) * 64 + lane;
                const int rowall = s >> 6, d = rowall / (2 * RI), orow = rowall - d * 2 * RI;
                const int ox = (s >> 1) & 31, half = s & 1;
                dpos[n] = ox | (orow << 8);
                dbase[n] = i < NI ? (unsigned)((((d * Ho + orow) * Wo + ox) * 8 + half * 4) * 4) : 0x80000000u;
            }
        }
        auto dma_tile = [&](unsigned t, int stage, bool live) {
            int b, i0, j0;
            decode(t, b, i0, j0);
            const unsigned porigin = (unsigned)((((b * D * a.Hi + i0) * a.Wi) + j0) * 64);
            const unsigned sorigin = (unsigned)((((b * D * Ho + 2 * i0) * Wo) + 2 * j0) * 32);
            const unsigned wlim = live ? (unsigned)a.Wi : 0u;
            f32x4v* const dst0 = lds + stage * G::STAGE;
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const int i = wave + 4 * n;
                const bool sk = i >= NBLK;
                const int cx = dpos[n] & 255, cy = dpos[n] >> 8;
                const bool ok = sk ? (2 * i0 + cy < Ho && (unsigned)(2 * j0 + cx) < 2 * wlim) : (i0 + cy < a.Hi && (unsigned)(j0 + cx) < wlim);
                const unsigned off = ok ? dbase[n] + (sk ? sorigin : porigin) : 0x80000000u;
                f32x4v* const dst = i < NI ? dst0 + i * 64 : scratch;
                if (sk) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(skip_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                }
            }
        };
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
    // weight fragments [dy][channel quad s]: row lm = (dx, co) of class q = 2 dy + dx, K slot (s, j, lq) = channel 4 s + j of
    // neighbourhood block lq = (di, dj)
    f32x4v wf[2][4];
    {
        const int dx = lm >> 3, co = lm & 7, di = lq >> 1, dj = lq & 1;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int ky = dy + 1 - 2 * di, kx = dx + 1 - 2 * dj;
            const bool nz = ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = a.w[nz ? ((ky * 3 + kx) * 16 + 4 * s + j) * 8 + co : 0];
                    wf[dy][s][j] = nz ? v : 0.0f;
                }
        }
    }
    const int c0 = (lq & 1) * 4;                               // this lane's channel quad of the 8 output channels
    const f32x4v scv = *reinterpret_cast<const f32x4v*>(a.scale + c0), shv = *reinterpret_cast<const f32x4v*>(a.shift + c0);
    const f32x4v pwv = *reinterpret_cast<const f32x4v*>(a.prob_w + c0);
    const float pb = a.prob_b[0];
    // operand read: voxel (row + di, lm + dj) of the slice's patch, channel quad s at slot quad s ^ ((v >> 2) & 3)
    const int vpos = lm + (lq & 1);
    const int abase = ((lq >> 1) * 17 + vpos) * 4, aswz = (vpos >> 2) & 3;
    // selection: thread t of the 256 compute threads owns output pixel (t >> 5, t & 31) of the 2 RI x 32 tile
    const int tsel = wave * 64 + lane, sy = tsel >> 5, sx = tsel & 31;
    const long hw = (long)Ho * Wo;
    const bool has_skip = a.skip != nullptr;

    __builtin_amdgcn_s_barrier();                               // the first stage has landed
    int st = 0, it = 0;
    for (; tile < a.ntiles; tile += nwg, ++it) {
        int b, i0, j0;
        decode(tile, b, i0, j0);
        const f32x4v* const stage = lds + st * G::STAGE;
        st = st + 1 == R ? 0 : st + 1;
        float* const lg = lgbuf + (it & 1) * G::LG;
        // this thread's output pixel and its hypotheses (in flight under the MFMA phase)
        const int oy = 2 * i0 + sy, ox = 2 * j0 + sx;
        const bool sel = tsel < 64 * RI && oy < Ho && ox < Wo;
        const long p = (long)oy * Wo + ox, vol = (long)b * D * hw;
        float hv[mv::kSelMaxD];
#pragma unroll
        for (int d = 0; d < D; ++d) hv[d] = sel ? a.hypo[vol + d * hw + p] : 1.0f;

#pragma unroll
        for (int sl = 0; sl < SPW; ++sl) {
            const int d = wave + 4 * sl;
            const f32x4v* const patch = stage + d * PSL;
#pragma unroll
            for (int r = 0; r < RI; ++r) {
                f32x4v X[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) X[s] = patch[r * 68 + abase + (s ^ aswz)];
                f32x4v sk[2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
                    sk[dy] = has_skip ? stage[NBLK * 64 + (d * 2 * RI + 2 * r + dy) * 64 + 4 * lm + lq] : (f32x4v){0.f, 0.f, 0.f, 0.f};
                f32x4v acc[2] = {(f32x4v){0.f, 0.f, 0.f, 0.f}, (f32x4v){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int dy = 0; dy < 2; ++dy)
                            acc[dy] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[dy][s][j], X[s][j], acc[dy], 0, 0, 0);
                // lane (lm, lq): channels c0 .. c0 + 3 of output pixel (2 (i0 + r) + dy, 2 (j0 + lm) + (lq >> 1))
#pragma unroll
                for (int dy = 0; dy < 2; ++dy) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[k] = fmaf(acc[dy][k], scv[k], shv[k]);
                        if (a.relu) v[k] = fmaxf(v[k], 0.0f);
                        if (has_skip) v[k] += sk[dy][k];
                    }
                    // `prob` head: lo = channels 0..3, hi = channels 4..7, logit = (lo + hi) + b  (conv_small.hip prob_logit)
                    float part = v[0] * pwv[0];
#pragma unroll
                    for (int k = 1; k < 4; ++k) part = fmaf(v[k], pwv[k], part);
                    const float other = __shfl_xor(part, 16);
                    const float logit = ((lq & 1) ? other + part : part + other) + pb;
                    if (!(lq & 1)) lg[(d * 2 * RI + 2 * r + dy) * 32 + 2 * lm + (lq >> 1)] = logit;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                     // LDS reads of this stage done, logits written
        __builtin_amdgcn_s_barrier();
        if (sel) {
            float l[mv::kSelMaxD];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                l[d] = lg[(d * 2 * RI + sy) * 32 + sx];
                if (a.logits_out) a.logits_out[vol + d * hw + p] = l[d];
            }
            const long img = (long)b * hw;
            mv::select_from_logits_vals(l, hv, a.attn + vol, a.depth + img, a.conf ? a.conf + img : nullptr,
                                        a.inv_min ? a.inv_min + img : nullptr, a.inv_max ? a.inv_max + img : nullptr, D, hw, p,
                                        a.split_itv);
        }
    }
}

template <int D, int RI, int R>
int launch_dsel(DselArgs& a, hipStream_t s) {
    using G = DselGeom<D, RI, R>;
    auto kern = deconv_select_mfma_kernel<D, RI, R>;
    static unsigned long attr_done = 0;
    if (G::LDS > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    if (G::LDS > 64 * 1024 && !mvconv::allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = mvconv::num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    const unsigned tx = (unsigned)((a.Wi + 15) / 16), ty = (unsigned)((a.Hi + RI - 1) / RI);
    const long ntiles = (long)tx * ty * a.B;
    if (ntiles >= (1L << 30)) return MVSTER_ERR_SHAPE;
    a.ntiles = (unsigned)ntiles;
    a.tiles_x = mv_fastdiv(tx);
    a.tiles_y = mv_fastdiv(ty);
    int per_cu = (int)((160 * 1024) / G::LDS);
    if (per_cu > 2) per_cu = 2;
    const long gmax = (long)ncu * per_cu;
    const long rounds = (ntiles + gmax - 1) / gmax;
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("deconv_select_mfma_kernel<%d, %d, %d>", D, RI, R);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), G::LDS, s, a);
    return mv_check_launch();
}

}  // namespace

namespace mvconv {

// Called by mvster_deconv_select (conv_small.hip) for D in {4, 8}; MVSTER_ERR_UNSUPPORTED = not covered (the VALU kernel runs).
int dispatch_deconv_select_mfma(const float* in, const float* w, const float* scale, const float* shift, const float* skip,
                                const float* prob_w, const float* prob_b, const float* hypo, float* attn, float* depth,
                                float* conf, float* inv_min, float* inv_max, float* logits_out, int B, int D, int Hi, int Wi,
                                int relu, float split_itv, hipStream_t s) {
    const long in_bytes = (long)B * D * Hi * Wi * 64, skip_bytes = (long)B * D * Hi * Wi * 4 * 32;
    if (in_bytes >= (1L << 31) || skip_bytes >= (1L << 31)) return MVSTER_ERR_UNSUPPORTED;
    DselArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.skip = skip; a.prob_w = prob_w; a.prob_b = prob_b; a.hypo = hypo;
    a.attn = attn; a.depth = depth; a.conf = conf; a.inv_min = inv_min; a.inv_max = inv_max; a.logits_out = logits_out;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.relu = relu; a.split_itv = split_itv;
    a.in_bytes = (unsigned)in_bytes; a.skip_bytes = (unsigned)skip_bytes;
    const int ncu = num_cus();
    if (D == 4) {
        // two-row tiles.  Large maps: two ring stages = 63 KB of LDS = two workgroups per CU (25.3 us at stage 4, 119 us at
        // 4 x 576 x 800; four-row tiles with one 120 KB workgroup per CU measured 27.1 / 134.6 us); small maps: three stages, one
        // workgroup per CU (stage 3: 10.9 against 11.8 us)
        const long t4 = (long)((Wi + 15) / 16) * ((Hi + 3) / 4) * B;
        return t4 >= 2L * ncu ? launch_dsel<4, 2, 2>(a, s) : launch_dsel<4, 2, 3>(a, s);
    }
    if (D == 8) {
        // (the coarse stages' maps: one-row tiles while two-row tiles would leave CUs without one)
        const long t2 = (long)((Wi + 15) / 16) * ((Hi + 1) / 2) * B;
        return t2 >= ncu ? launch_dsel<8, 2, 2>(a, s) : launch_dsel<8, 1, 2>(a, s);
    }
    return MVSTER_ERR_UNSUPPORTED;
}

}  // namespace mvconv
