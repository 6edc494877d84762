// 3x3 (and 3x3x3) stride-1 convolutions of the 16 / 32 / 64-channel layers with fp32 PRODUCTS ON THE BF16 MATRIX CORES
// (round 5 probe, variant 11 of mvster_conv_mfma).
//
// Why: on gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD, 157 TFLOP/s): the dense layers of the
// forward (78 GFLOP) cannot take less than ~0.5 ms on it, and the Winograd kernels that carry most of them sit at 0.23-0.29
// of that peak because their transforms share the same VALU budget.  The bf16 MFMA is 16x faster.  An fp32 number splits
// EXACTLY into three bf16 numbers (24 significand bits = 8 + 8 + 8, round-to-nearest at every level, residuals computed
// exactly in fp32): x = x1 + x2 + x3.  A product a*b then is the sum of nine bf16 x bf16 products, each exact in fp32; the
// three smallest (a2 b3, a3 b2, a3 b3 <= 2^-24 |ab|, random sign) are dropped, the other six run as six bf16 MFMAs with fp32
// accumulation: 6/16 = 0.375 of the fp32-MFMA time at the accuracy of an fp32 dot product (dropped terms ~2^-24 relative
// per product, i.e. one fp32 rounding; the big term a1 b1 and the five corrections accumulate in separate registers so the
// corrections are not rounded at the magnitude of the running sum).
//
// Frame (v2, persistent): workgroup = 512 threads = 4 compute waves + 4 loading waves, alive over its share of the work items
// (item = TY x 32 output pixels x NW = 16 NTW output channels of one (b, z) slice; TY = 4 TYQ rows, compute wave w owns TYQ
// rows).  The K loop of an item runs over stages = (kd, 16 input channels); the stages of all items of a workgroup form ONE
// stream that the loading waves run two stages ahead of the compute waves: global loads of stage k + 2 in flight (two
// register sets), stage k + 1 being split into three bf16 planes and written to the other LDS buffer, stage k under the
// MFMAs -- one barrier per stage.  (v1, one tile per workgroup and everything in sequence, measured 1.0x the Winograd
// kernels: a wave was alive for ~6 us around 0.4 us of matrix work, profiles/r05_b3_v1_check.txt.)
// A stage in LDS: the input patch ((TY + 2) x 34 pixels x 16 channels) as [plane][8-channel half][row][column] x 16 B, so a
// 16-lane fragment read is 256 contiguous bytes, and -- unless the layer's weights live in registers (WREG: the one-stage
// 16 -> 16 layers, 60 registers) -- the stage's weights, which arrive pre-split in fragment order (conv_plan.py:pack_b3).
// MFMA shape: v_mfma_f32_16x16x32_bf16 with A = weights (M = 16 output channels), B = pixels (N = 16 pixels of a row),
// K = 32 = two taps x 16 channels (the nine taps of a slice pair up as (0,1) (2,3) (4,5) (6,7) (8, zero)); lane (p = lane &
// 15, g = lane >> 4) holds 8 consecutive channels (half g & 1) of tap 2 tp + (g >> 1) -- the same rule on both operands, so
// no K permutation exists.  D layout: column = lane & 15 = pixel, row = 4 g + reg = output channel: each lane ends up with
// four consecutive output channels of one pixel = one 16-byte store.
// Epilogue as in every other kernel of the family: y = acc * scale[co] + shift[co], ReLU, + skip (same resolution).
// Reference layers: models/mvs4net_utils.py:430-446 (FPN conv1-3 3x3 layers), :457-459 (out2 / out3), :877-883 (reg2d
// conv2 / conv4 / conv6).
#include "conv_args.hpp"

namespace {

using mvconv::ConvArgs;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4b __attribute__((ext_vector_type(4)));

struct B3Args {
    const float* in;      // [B, D, H, W, CIN]
    const bf16x8* w;      // [KD][CIN/16][nsplit][1024 * NTW] 16-byte units (fragment order, zero-padded to 16 KB per N tile)
    const float* scale;   // [COUT]
    const float* shift;   // [COUT]
    const float* skip;    // [B, D, H, W, COUT] or null
    float* out;           // [B, D, H, W, COUT]
    int B, D, H, W, cout, relu;
    unsigned tiles_x, tiles_y, nsplit, ntiles;
};

template <int TYQ>
struct B3Geom {
    static constexpr int TY = 4 * TYQ, PH = TY + 2, PW = 34;
    static constexpr int PLANE = 2 * PH * PW;                 // 16-byte units of one bf16 plane (two 8-channel halves)
    static constexpr int PATCH = 3 * PLANE;
    static constexpr int UNITS = 2 * PH * PW;                 // 32-byte fp32 units (pixel, half) a stage loads
    static constexpr int NU = (UNITS + 255) / 256;            // per thread
};

// x -> (bf16(x), x - bf16(x)): round-to-nearest-even conversion, exact fp32 residual
__device__ __forceinline__ void split_pair(float a, float b, bf16x2& p, float& ra, float& rb) {
    p[0] = (__bf16)a;
    p[1] = (__bf16)b;
    ra = a - (float)p[0];
    rb = b - (float)p[1];
}

__device__ __forceinline__ void split8(const f32x4b lo, const f32x4b hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    float e[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x2 a, b, c;
        float r0, r1, s0, s1, t0, t1;
        split_pair(e[2 * i], e[2 * i + 1], a, r0, r1);
        split_pair(r0, r1, b, s0, s1);
        split_pair(s0, s1, c, t0, t1);
        p1[2 * i] = a[0]; p1[2 * i + 1] = a[1];
        p2[2 * i] = b[0]; p2[2 * i + 1] = b[1];
        p3[2 * i] = c[0]; p3[2 * i + 1] = c[1];
    }
}

struct B3Item { int ns, b, z, y0, x0, s_begin, s_end; };

template <int CIN, int NTW, int KD, int TYQ, bool WREG>
__global__ void __launch_bounds__(512) conv_b3_kernel(B3Args a) {
    using G = B3Geom<TYQ>;
    constexpr int PH = G::PH, PW = G::PW, PLANE = G::PLANE, NU = G::NU, NCH = CIN / 16;
    constexpr int PT = 2 * TYQ;                                // 16-pixel tiles per compute wave
    constexpr int WUNITS = 1024 * NTW;                         // 16-byte units of a stage's weight block (padded)
    constexpr int NWL = WREG ? 0 : WUNITS / 256;               // weight units per loading thread and stage
    constexpr int BUFU = G::PATCH + (WREG ? 0 : WUNITS);       // 16-byte units of one stage buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8* const lds = reinterpret_cast<bf16x8*>(smem);       // two stage buffers: [patch 3 x 2 x PH x PW][weights 5 x NTW x 3 x 64 + pad]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const unsigned nwg = gridDim.x, nitems = a.ntiles * a.nsplit;
    const unsigned first = xcd_remap(blockIdx.x, nwg);
    if (first >= nitems) return;

    auto decode = [&](unsigned item) {
        B3Item it;
        it.ns = (int)(item % a.nsplit);
        unsigned t = item / a.nsplit;
        const int tx = (int)(t % a.tiles_x); t /= a.tiles_x;
        const int ty = (int)(t % a.tiles_y); t /= a.tiles_y;
        it.z = (int)(t % (unsigned)a.D);
        it.b = (int)(t / (unsigned)a.D);
        it.y0 = ty * G::TY;
        it.x0 = tx * 32;
        // stages: kd in [kd_lo, kd_hi] (the slices inside the volume) x the NCH channel chunks
        const int kd_lo = KD == 1 ? 0 : (it.z == 0 ? 1 : 0);
        const int kd_hi = KD == 1 ? 0 : (it.z == a.D - 1 ? KD - 2 : KD - 1);
        it.s_begin = kd_lo * NCH;
        it.s_end = (kd_hi + 1) * NCH;
        return it;
    };

    if (loader) {
        // ------------------------------------------------------------------------------------------------ loading waves
        const int ltid = tid & 255;
        // this thread's patch units: (row r, column c, half c8) -> one 32-byte global run, three 16-byte LDS slots
        int unit_lds[NU], unit_r[NU], unit_c[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int u = ltid + 256 * i;
            const int c8 = u & 1, pix = u >> 1;
            unit_r[i] = pix / PW;
            unit_c[i] = pix - unit_r[i] * PW;
            unit_lds[i] = u < G::UNITS ? (c8 * PH + unit_r[i]) * PW + unit_c[i] : -1;
        }
        struct Raw { f32x4b v[NU][2]; bf16x8 w[NWL > 0 ? NWL : 1]; };
        auto issue = [&](const B3Item& it, int s, Raw& r) {
            const int kd = s / NCH, ch = s - kd * NCH;
            const int dz = it.z + kd - KD / 2;
            const float* slice = a.in + ((long)(it.b * a.D + dz) * a.H * a.W) * CIN + ch * 16 + (ltid & 1) * 8;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int y = it.y0 - 1 + unit_r[i], x = it.x0 - 1 + unit_c[i];
                if (unit_lds[i] >= 0 && y >= 0 && y < a.H && x >= 0 && x < a.W) {
                    const float* src = slice + ((long)y * a.W + x) * CIN;
                    r.v[i][0] = *reinterpret_cast<const f32x4b*>(src);
                    r.v[i][1] = *reinterpret_cast<const f32x4b*>(src + 4);
                } else {
                    r.v[i][0] = f32x4b{0.f, 0.f, 0.f, 0.f};
                    r.v[i][1] = f32x4b{0.f, 0.f, 0.f, 0.f};
                }
            }
            if constexpr (!WREG) {
                const bf16x8* wsrc = a.w + ((long)s * a.nsplit + it.ns) * WUNITS;
#pragma unroll
                for (int i = 0; i < NWL; ++i) r.w[i] = wsrc[ltid + 256 * i];
            }
        };
        auto write = [&](const Raw& r, bf16x8* buf) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                if (unit_lds[i] >= 0) {
                    bf16x8 p1, p2, p3;
                    split8(r.v[i][0], r.v[i][1], p1, p2, p3);
                    buf[unit_lds[i]] = p1;
                    buf[PLANE + unit_lds[i]] = p2;
                    buf[2 * PLANE + unit_lds[i]] = p3;
                }
            }
            if constexpr (!WREG) {
#pragma unroll
                for (int i = 0; i < NWL; ++i) buf[G::PATCH + ltid + 256 * i] = r.w[i];
            }
        };
        // cursor over the stage stream of this workgroup, two stages ahead of the one being written
        unsigned item = first;
        B3Item it = decode(item);
        int s = it.s_begin;
        bool live = true;
        auto advance = [&]() {
            if (++s >= it.s_end) {
                item += nwg;
                live = item < nitems;
                if (live) {
                    it = decode(item);
                    s = it.s_begin;
                }
            }
        };
        Raw ra, rb;
        issue(it, s, ra);                                      // stage 0 -> set A
        advance();
        bool have_b = live;
        if (have_b) {
            issue(it, s, rb);                                  // stage 1 -> set B
            advance();
        }
        int k = 0;
        while (true) {
            write(ra, lds + (k & 1) * BUFU);                   // stage k (set A)
            const bool more_a = have_b && live;                // is there a stage k + 2?
            if (more_a) {
                issue(it, s, ra);
                advance();
            }
            __syncthreads();
            ++k;
            if (!have_b) break;
            write(rb, lds + (k & 1) * BUFU);                   // stage k + 1 (set B)
            have_b = more_a && live;                           // is there a stage k + 3?
            if (have_b) {
                issue(it, s, rb);
                advance();
            }
            __syncthreads();
            ++k;
            if (!more_a) break;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- compute waves
    const int p = lane & 15, g = lane >> 4;
    // fragment addresses: lane (p, g) reads channels half g & 1 of tap 2 tp + (g >> 1) (tap 9 does not exist: its weights
    // are zero, the pixel operand re-reads tap 8 so that the product is 0 x (a value of this output's own footprint))
    int poff[5];
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
        int tap = 2 * tp + (g >> 1);
        tap = tap > 8 ? 8 : tap;
        const int ky = tap / 3, kx = tap - 3 * ky;
        poff[tp] = ((g & 1) * PH + ky + wave * TYQ) * PW + kx + p;
    }
    bf16x8 wreg[WREG ? 5 : 1][NTW][3];
    if constexpr (WREG) {
#pragma unroll
        for (int tp = 0; tp < 5; ++tp)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wreg[tp][j][pl] = a.w[((tp * NTW + j) * 3 + pl) * 64 + lane];
    }

    f32x4b acc_hi[PT][NTW], acc_lo[PT][NTW];
    int k = 0;
    for (unsigned item = first; item < nitems; item += nwg) {
        const B3Item it = decode(item);
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc_hi[i][j] = f32x4b{0.f, 0.f, 0.f, 0.f};
                acc_lo[i][j] = f32x4b{0.f, 0.f, 0.f, 0.f};
            }
        for (int s = it.s_begin; s < it.s_end; ++s, ++k) {
            __syncthreads();                                   // stage k has landed in buffer k & 1
            const bf16x8* patch = lds + (k & 1) * BUFU;
            const bf16x8* wl = patch + G::PATCH;
#pragma unroll
            for (int tp = 0; tp < 5; ++tp) {
                bf16x8 wf[NTW][3];
#pragma unroll
                for (int j = 0; j < NTW; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        if constexpr (WREG) wf[j][pl] = wreg[tp][j][pl];
                        else wf[j][pl] = wl[((tp * NTW + j) * 3 + pl) * 64 + lane];
                    }
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    bf16x8 pf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) pf[pl] = patch[pl * PLANE + poff[tp] + (i >> 1) * PW + (i & 1) * 16];
#pragma unroll
                    for (int j = 0; j < NTW; ++j) {
                        // corrections first (smallest first), the leading term into its own accumulator
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][2], pf[0], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], pf[2], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], pf[1], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], pf[0], acc_lo[i][j], 0, 0, 0);
                        acc_lo[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], pf[1], acc_lo[i][j], 0, 0, 0);
                        acc_hi[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], pf[0], acc_hi[i][j], 0, 0, 0);
                    }
                }
            }
        }
        // epilogue: lane = (pixel p of the 16-pixel tile, output channels 4 g .. 4 g + 3 of the N tile)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int co = (it.ns * NTW + j) * 16 + 4 * g;
            const f32x4b sc = *reinterpret_cast<const f32x4b*>(a.scale + co);
            const f32x4b sh = *reinterpret_cast<const f32x4b*>(a.shift + co);
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int y = it.y0 + wave * TYQ + (i >> 1), x = it.x0 + (i & 1) * 16 + p;
                if (y < a.H && x < a.W) {
                    const long o = ((((long)it.b * a.D + it.z) * a.H + y) * a.W + x) * a.cout + co;
                    f32x4b v = (acc_hi[i][j] + acc_lo[i][j]) * sc + sh;
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    if (a.skip) v += *reinterpret_cast<const f32x4b*>(a.skip + o);
                    *reinterpret_cast<f32x4b*>(a.out + o) = v;
                }
            }
        }
    }
}

template <int CIN, int NTW, int KD, int TYQ, bool WREG>
int launch_b3(const B3Args& a, int wpc, hipStream_t s) {
    using G = B3Geom<TYQ>;
    const size_t lds = (size_t)2 * (G::PATCH + (WREG ? 0 : 1024 * NTW)) * 16;
    static unsigned long big_done = 0;
    auto kern = conv_b3_kernel<CIN, NTW, KD, TYQ, WREG>;
    if (lds > 64 * 1024 && !mvconv::allow_big_lds(reinterpret_cast<const void*>(kern), big_done)) return MVSTER_ERR_LAUNCH;
    const int cus = mvconv::num_cus();
    if (cus <= 0) return MVSTER_ERR_LAUNCH;
    const unsigned fit = (unsigned)((160 * 1024) / lds);                     // workgroups one CU's LDS holds
    unsigned per_cu = wpc > 0 ? (unsigned)wpc : 1u;
    per_cu = per_cu > fit ? fit : per_cu;
    per_cu = per_cu > 2 ? 2 : per_cu;                                        // (512 threads x 2 = the CU's 16 wave slots at <= 128 VGPRs)
    const unsigned nitems = a.ntiles * a.nsplit;
    const unsigned grid = nitems < (unsigned)cus * per_cu ? nitems : (unsigned)cus * per_cu;
    MV_NOTE_KERNEL("conv_b3_kernel<%d, %d, %d, %d, %s>", CIN, NTW, KD, TYQ, WREG ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    return mv_check_launch();
}

template <int CIN, int NTW>
int launch_b3_shape(const B3Args& a, int kd, int tyq, int wpc, hipStream_t s) {
    if (kd == 1) {
        if constexpr (CIN == 16 && NTW == 1)                  // one stage per item: the weights stay in registers
            return tyq == 1 ? launch_b3<CIN, NTW, 1, 1, true>(a, wpc, s) : launch_b3<CIN, NTW, 1, 2, true>(a, wpc, s);
        else
            return tyq == 1 ? launch_b3<CIN, NTW, 1, 1, false>(a, wpc, s) : launch_b3<CIN, NTW, 1, 2, false>(a, wpc, s);
    }
    return tyq == 1 ? launch_b3<CIN, NTW, 3, 1, false>(a, wpc, s) : launch_b3<CIN, NTW, 3, 2, false>(a, wpc, s);
}

}  // namespace

namespace mvconv {

// variant 11 of mvster_conv_mfma; `wpk` = the pre-split bf16 fragments of conv_plan.py:pack_b3; mt = rows per compute wave
// (1 | 2); wpc = workgroups per CU (0 = 1)
int dispatch_b3(const ConvArgs& c, int mt, int wpc, hipStream_t s) {
    if (c.nclass != 1 || c.sd != 1 || c.sh != 1 || c.sw != 1 || c.osd != 1 || c.osh != 1 || c.osw != 1) return MVSTER_ERR_UNSUPPORTED;
    const int kd = c.kd[0];
    if (c.kh[0] != 3 || c.kw[0] != 3 || (kd != 1 && kd != 3) || c.ph[0] != 1 || c.pw[0] != 1 || c.pd[0] != kd / 2)
        return MVSTER_ERR_UNSUPPORTED;
    if (c.Do != c.Di || c.Ho != c.Hi || c.Wo != c.Wi || c.prob_w || c.skip_mode == 2) return MVSTER_ERR_UNSUPPORTED;
    if (c.cout != 16 && c.cout != 32 && c.cout != 64) return MVSTER_ERR_UNSUPPORTED;
    if ((long)c.B * c.Do * c.Ho * c.Wo * c.cout >= (1L << 31)) return MVSTER_ERR_SHAPE;
    const int ntw = c.cout == 16 ? 1 : 2;
    const int tyq = mt == 1 ? 1 : 2;
    B3Args a;
    a.in = c.in; a.w = reinterpret_cast<const bf16x8*>(c.wpk); a.scale = c.scale; a.shift = c.shift;
    a.skip = c.skip_mode == 1 ? c.skip : nullptr; a.out = c.out;
    a.B = c.B; a.D = c.Do; a.H = c.Ho; a.W = c.Wo; a.cout = c.cout; a.relu = c.relu;
    a.tiles_x = (unsigned)((c.Wo + 31) / 32);
    a.tiles_y = (unsigned)((c.Ho + 4 * tyq - 1) / (4 * tyq));
    a.nsplit = (unsigned)(c.cout / (16 * ntw));
    const long ntiles = (long)a.tiles_x * a.tiles_y * c.Do * c.B;
    if (ntiles * a.nsplit >= (1L << 31)) return MVSTER_ERR_SHAPE;
    a.ntiles = (unsigned)ntiles;
    switch (c.cin) {
        case 16: return ntw == 1 ? launch_b3_shape<16, 1>(a, kd, tyq, wpc, s) : launch_b3_shape<16, 2>(a, kd, tyq, wpc, s);
        case 32: return ntw == 1 ? launch_b3_shape<32, 1>(a, kd, tyq, wpc, s) : launch_b3_shape<32, 2>(a, kd, tyq, wpc, s);
        case 64: return ntw == 1 ? launch_b3_shape<64, 1>(a, kd, tyq, wpc, s) : launch_b3_shape<64, 2>(a, kd, tyq, wpc, s);
        default: return MVSTER_ERR_UNSUPPORTED;
    }
}

}  // namespace mvconv
