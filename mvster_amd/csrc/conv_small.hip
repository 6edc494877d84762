// 3x3 convolution for the narrow full-resolution layers (Cin in {4, 8}, Cout = 8) on the fp32 VALU.
//
// These layers (FPN conv0.*, the composed FPN tail conv, conv0 of every reg2d) have K = 36 or 72 and
// N = 8: on the 16x16x4 MFMA half of every N tile is padding and the K loop is 3-5 steps long, so the
// matrix path tops out near 25 TFLOP/s there.  Here one thread owns one output voxel and its 8 output
// channels (8 packed-FMA accumulators), the 10 x 34 input patch of an 8 x 32 tile is staged once in
// LDS (zero padding via the buffer range check), and the 9 x Cin x 8 weights are wave-uniform scalar
// loads -- no padding waste, coalesced 32-byte stores.  fp32 FMA chain in (tap, cin) order.
// Reference layers: models/mvs4net_utils.py:427-428 (FPN conv0), :875 (reg2d conv0), :459 (out4, composed).
#include "conv_args.hpp"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct SmallArgs {
    const float* in;     // [NB, H, W, CIN]
    const float* w;      // [9][CIN][8]
    const float* scale;  // [8]
    const float* shift;  // [8]
    const float* skip;   // [NB, H, W, 8] or null
    float* out;          // [NB, H, W, 8]
    int NB, H, W, relu;
    unsigned in_bytes;
};

template <int CIN>
__global__ void __launch_bounds__(256) conv_small_kernel(SmallArgs a, FastDiv tiles_x, FastDiv tiles_y) {
    constexpr int TY = 8, TX = 32, PH = TY + 2, PW = TX + 2, Q = CIN / 4;
    __shared__ f32x4v patch[PH * PW * Q];
    unsigned txu, tyu;
    const int nb = (int)fdivmod(fdivmod(xcd_remap(blockIdx.x, gridDim.x), tiles_x, txu), tiles_y, tyu);
    const int y0 = (int)tyu * TY, x0 = (int)txu * TX;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;

    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
    constexpr int NST = (PH * PW * Q + 255) / 256;
    f32x4v tmp[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int q = idx % Q, pix = idx / Q;
        const int px = pix % PW, py = pix / PW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = idx < PH * PW * Q && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const unsigned off = ok ? (unsigned)(((nb * a.H + iy) * a.W + ix) * CIN + q * 4) * 4u : 0xFFFFFFF0u;
        tmp[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int idx = threadIdx.x + i * 256;
        // plane layout [q][row][col]: neighbouring pixels are 16 bytes apart for the tap reads (conflict-free b128)
        if (idx < PH * PW * Q) patch[(idx % Q) * (PH * PW) + idx / Q] = tmp[i];
    }
    __syncthreads();

    f32x2v acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x2v){0.f, 0.f};
    const f32x2v* w2 = reinterpret_cast<const f32x2v*>(a.w);   // wave-uniform -> scalar loads
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f32x4v* p = patch + (ty + ky) * PW + tx + kx;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const f32x4v xv = p[q * (PH * PW)];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ci = q * 4 + c;
                    const f32x2v xx = {xv[c], xv[c]};
                    const f32x2v* wr = w2 + ((ky * 3 + kx) * CIN + ci) * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_elementwise_fma(xx, wr[j], acc[j]);
                }
            }
        }

    const int y = y0 + ty, x = x0 + tx;
    if (y >= a.H || x >= a.W) return;
    const long o = (((long)nb * a.H + y) * a.W + x) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = acc[j][0]; v[2 * j + 1] = acc[j][1]; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        v[c] = fmaf(v[c], a.scale[c], a.shift[c]);
        if (a.relu) v[c] = fmaxf(v[c], 0.0f);
    }
    if (a.skip) {
        const f32x4v s0 = ld4(a.skip + o), s1 = ld4(a.skip + o + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { v[c] += s0[c]; v[4 + c] += s1[c]; }
    }
    st4(a.out + o, (f32x4v){v[0], v[1], v[2], v[3]});
    st4(a.out + o + 4, (f32x4v){v[4], v[5], v[6], v[7]});
}

}  // namespace

extern "C" int mvster_conv_small(const float* in, const float* w, const float* scale, const float* shift,
                                 const float* skip, float* out, int NB, int H, int W, int cin, int relu,
                                 void* stream) {
    if (!in || !w || !scale || !shift || !out) return MVSTER_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0) return MVSTER_ERR_SHAPE;
    const long in_elems = (long)NB * H * W * cin;
    if (in_elems >= (1L << 30)) return MVSTER_ERR_SHAPE;
    SmallArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.skip = skip; a.out = out;
    a.NB = NB; a.H = H; a.W = W; a.relu = relu; a.in_bytes = (unsigned)(in_elems * 4);
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
    const long blocks = (long)tiles_x * tiles_y * NB;
    if (blocks >= (1L << 31)) return MVSTER_ERR_SHAPE;
    dim3 grid((unsigned)blocks), block(256);
    if (cin == 8) {
        MV_NOTE_KERNEL("conv_small_kernel<8>");
        hipLaunchKernelGGL(conv_small_kernel<8>, grid, block, 0, (hipStream_t)stream, a, mv_fastdiv(tiles_x), mv_fastdiv(tiles_y));
    } else if (cin == 4) {
        MV_NOTE_KERNEL("conv_small_kernel<4>");
        hipLaunchKernelGGL(conv_small_kernel<4>, grid, block, 0, (hipStream_t)stream, a, mv_fastdiv(tiles_x), mv_fastdiv(tiles_y));
    } else {
        return MVSTER_ERR_UNSUPPORTED;
    }
    return mv_check_launch();
}

// ------------------------------------------------------------------------------------------
// Transposed (1,3,3) stride-2 convolution (pad 1, output_padding 1) for the two finest up-sampling
// layers of reg2d (conv9: 32 -> 16, conv11: 16 -> 8; models/mvs4net_utils.py:890-898) on the VALU.
// One thread owns one INPUT lattice voxel (i, j) and produces its 2 x 2 output block:
//   out[2i  ][2j  ] = x(i,j) W[1][1]
//   out[2i  ][2j+1] = x(i,j) W[1][2] + x(i,j+1) W[1][0]
//   out[2i+1][2j  ] = x(i,j) W[2][1] + x(i+1,j) W[0][1]
//   out[2i+1][2j+1] = x(i,j) W[2][2] + x(i,j+1) W[2][0] + x(i+1,j) W[0][2] + x(i+1,j+1) W[0][0]
// (o = 2 i - 1 + k).  The layers are HBM-bound (the skip tensor and the output are 2-4x the input);
// weights are wave-uniform scalar loads, stores are 64 contiguous bytes per thread and row.  Epilogue:
// BatchNorm scale/shift, ReLU, skip add, optionally the fused 1x1x1 `prob` head (COUT == 8).
// ------------------------------------------------------------------------------------------
namespace {

struct DeconvArgs {
    const float* in;      // [NB, Hi, Wi, CIN]
    const float* w;       // [3][3][CIN][COUT]
    const float* scale;
    const float* shift;
    const float* skip;    // [NB, 2Hi, 2Wi, COUT] or null
    const float* prob_w;  // optional [COUT] (COUT == 8)
    const float* prob_b;
    float* out;           // [NB, 2Hi, 2Wi, COUT]  or logits [NB, 2Hi, 2Wi]
    int NB, Hi, Wi, relu;
    unsigned in_bytes;
};

// The 2 x 2 outputs (all COUT channels, after scale / shift, ReLU and the skip connection) of input voxel (nb, i, j).
template <int CIN, int COUT>
__device__ __forceinline__ void deconv_voxel(const DeconvArgs& a, int nb, int i, int j, float (&v)[4][COUT]) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
    const unsigned base = (unsigned)((nb * a.Hi + i) * a.Wi + j) * CIN * 4u;
    const bool jn = j + 1 < a.Wi, in_ = i + 1 < a.Hi;
    const unsigned o01 = jn ? base + CIN * 4u : 0xFFFFFFF0u;
    const unsigned o10 = in_ ? base + (unsigned)a.Wi * CIN * 4u : 0xFFFFFFF0u;
    const unsigned o11 = (jn && in_) ? base + (unsigned)(a.Wi + 1) * CIN * 4u : 0xFFFFFFF0u;

    // packed fp32 FMAs (two output channels per instruction), weights as wave-uniform scalar pairs.  The channel loop
    // stays rolled (4 channels per trip): fully unrolled, the 1152 scalar weights and 576 packed FMAs of (16, 8) ran
    // 45.8 us at stage 4 against 24.3 us rolled (25.9 us with unpacked FMAs).
    f32x2v acc[4][COUT / 2];   // [dy*2+dx][co pair]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < COUT / 2; ++c) acc[q][c] = (f32x2v){0.f, 0.f};

    const f32x2v* w2 = reinterpret_cast<const f32x2v*>(a.w);
    auto wrow = [&](int ky, int kx, int ci) { return w2 + ((ky * 3 + kx) * CIN + ci) * (COUT / 2); };   // wave-uniform
    // the rolled channel loop is software-pipelined: the four corner loads of trip t + 1 are issued before the FMAs of
    // trip t (as plain load-then-compute trips every trip sat out a memory latency; fully unrolled, hipcc hoists all 1152
    // scalar weights at once and spills SGPRs into VGPR lanes)
    auto load4 = [&](int c4, f32x4v (&x)[4]) {
        const unsigned c4b = (unsigned)c4 * 4u;
        x[0] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + c4b, 0, 0));
        x[1] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o01 == 0xFFFFFFF0u ? o01 : o01 + c4b, 0, 0));
        x[2] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o10 == 0xFFFFFFF0u ? o10 : o10 + c4b, 0, 0));
        x[3] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o11 == 0xFFFFFFF0u ? o11 : o11 + c4b, 0, 0));
    };
    f32x4v xn[4];
    load4(0, xn);
#pragma unroll 1
    for (int c4 = 0; c4 < CIN; c4 += 4) {
        const f32x4v x00 = xn[0], x01 = xn[1], x10 = xn[2], x11 = xn[3];
        load4(c4 + 4 < CIN ? c4 + 4 : c4, xn);            // (unconditional: the last trip re-reads its own quad)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = c4 + k;
            const f32x2v a00 = {x00[k], x00[k]}, a01 = {x01[k], x01[k]}, a10 = {x10[k], x10[k]}, a11 = {x11[k], x11[k]};
#pragma unroll
            for (int co = 0; co < COUT / 2; ++co) {
                acc[0][co] = __builtin_elementwise_fma(a00, wrow(1, 1, ci)[co], acc[0][co]);
                acc[1][co] = __builtin_elementwise_fma(a00, wrow(1, 2, ci)[co], acc[1][co]);
                acc[1][co] = __builtin_elementwise_fma(a01, wrow(1, 0, ci)[co], acc[1][co]);
                acc[2][co] = __builtin_elementwise_fma(a00, wrow(2, 1, ci)[co], acc[2][co]);
                acc[2][co] = __builtin_elementwise_fma(a10, wrow(0, 1, ci)[co], acc[2][co]);
                acc[3][co] = __builtin_elementwise_fma(a00, wrow(2, 2, ci)[co], acc[3][co]);
                acc[3][co] = __builtin_elementwise_fma(a01, wrow(2, 0, ci)[co], acc[3][co]);
                acc[3][co] = __builtin_elementwise_fma(a10, wrow(0, 2, ci)[co], acc[3][co]);
                acc[3][co] = __builtin_elementwise_fma(a11, wrow(0, 0, ci)[co], acc[3][co]);
            }
        }
    }
    const int Ho = 2 * a.Hi, Wo = 2 * a.Wi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int oy = 2 * i + (q >> 1), ox = 2 * j + (q & 1);
        const long opix = ((long)nb * Ho + oy) * Wo + ox;
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            v[q][c] = fmaf(acc[q][c >> 1][c & 1], a.scale[c], a.shift[c]);
            if (a.relu) v[q][c] = fmaxf(v[q][c], 0.0f);
        }
        if (a.skip) {
#pragma unroll
            for (int c = 0; c < COUT; c += 4) {
                const f32x4v s4 = ld4(a.skip + opix * COUT + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[q][c + k] += s4[k];
            }
        }
    }
}

// 8-channel dot product of the `prob` head (reference reg2d.prob, mvs4net_utils.py:900), same association as everywhere
__device__ __forceinline__ float prob_logit(const float (&v)[8], const float* prob_w, const float* prob_b) {
    float lo = v[0] * prob_w[0], hi = v[4] * prob_w[4];
#pragma unroll
    for (int k = 1; k < 4; ++k) { lo = fmaf(v[k], prob_w[k], lo); hi = fmaf(v[4 + k], prob_w[4 + k], hi); }
    return (lo + hi) + prob_b[0];
}

template <int CIN, int COUT>
__global__ void __launch_bounds__(256) deconv_small_kernel(DeconvArgs a) {
    const int pix = xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int nb = blockIdx.y;
    if (pix >= a.Hi * a.Wi) return;
    const int i = pix / a.Wi, j = pix - i * a.Wi;
    float v[4][COUT];
    deconv_voxel<CIN, COUT>(a, nb, i, j, v);
    const int Ho = 2 * a.Hi, Wo = 2 * a.Wi;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int oy = 2 * i + (q >> 1), ox = 2 * j + (q & 1);
        const long opix = ((long)nb * Ho + oy) * Wo + ox;
        if constexpr (COUT == 8) {
            if (a.prob_w) {
                a.out[opix] = prob_logit(v[q], a.prob_w, a.prob_b);
                continue;
            }
        }
#pragma unroll
        for (int c = 0; c < COUT; c += 4) st4(a.out + opix * COUT + c, (f32x4v){v[q][c], v[q][c + 1], v[q][c + 2], v[q][c + 3]});
    }
}

// reg2d's last layer AND the depth selection in one launch: conv11 (ConvTranspose 16 -> 8, BatchNorm, ReLU, + skip c0),
// the 1x1x1 `prob` head, softmax over depth, first-max argmax, gather, confidence, inverse bounds
// (models/mvs4net_utils.py:897-900, :1068-1088).  The logits never leave the chip: workgroup = 64 input pixels x D
// hypotheses; thread (p, d) computes the 2 x 2 logits of its voxel, they meet in LDS, and thread (p, q) then finishes
// output pixel q of p's 2 x 2 block with mv::select_from_logits -- the arithmetic of deconv_small_kernel followed by
// select_depth_kernel, bit for bit.
struct SelectOut {
    const float* hypo;     // [B, D, Ho, Wo]
    float *attn, *depth, *conf, *inv_min, *inv_max, *logits_out;   // conf / inv_* / logits_out optional
    int D;
    float split_itv;
};

template <int CIN>
__global__ void __launch_bounds__(1024) deconv_select_kernel(DeconvArgs a, SelectOut so) {
    __shared__ float lg[mv::kSelMaxD][4][64];
    const int px = threadIdx.x, d = threadIdx.y;
    const int pix = xcd_remap(blockIdx.x, gridDim.x) * 64 + px;
    const int b = blockIdx.y;
    const bool inside = pix < a.Hi * a.Wi;
    const int pc = inside ? pix : a.Hi * a.Wi - 1;
    const int i = pc / a.Wi, j = pc - i * a.Wi;
    {
        float v[4][8];
        deconv_voxel<CIN, 8>(a, b * so.D + d, i, j, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) lg[d][q][px] = prob_logit(v[q], a.prob_w, a.prob_b);
    }
    __syncthreads();
    if (!inside) return;
    const int Ho = 2 * a.Hi, Wo = 2 * a.Wi;
    const long hw = (long)Ho * Wo;
    auto finish = [&](int q) {
        const long p = (long)(2 * i + (q >> 1)) * Wo + (2 * j + (q & 1));
        float l[mv::kSelMaxD];
#pragma unroll
        for (int dd = 0; dd < mv::kSelMaxD; ++dd) {
            if (dd >= so.D) break;
            l[dd] = lg[dd][q][px];
            if (so.logits_out) so.logits_out[((long)b * so.D + dd) * hw + p] = l[dd];
        }
        const long vol = (long)b * so.D * hw, img = (long)b * hw;
        mv::select_from_logits(l, so.hypo + vol, so.attn + vol, so.depth + img, so.conf ? so.conf + img : nullptr,
                               so.inv_min ? so.inv_min + img : nullptr, so.inv_max ? so.inv_max + img : nullptr, so.D, hw, p,
                               so.split_itv);
    };
    // output pixels q = d and d + D of the 2 x 2 block (D >= 2 covers all four)
    if (d < 4) finish(d);
    if (d + so.D < 4) finish(d + so.D);
}

}  // namespace

extern "C" int mvster_deconv_small(const float* in, const float* w, const float* scale, const float* shift,
                                   const float* skip, const float* prob_w, const float* prob_b, float* out, int NB,
                                   int Hi, int Wi, int cin, int cout, int relu, void* stream) {
    if (!in || !w || !scale || !shift || !out) return MVSTER_ERR_NULL;
    if ((prob_w == nullptr) != (prob_b == nullptr)) return MVSTER_ERR_NULL;
    if (NB <= 0 || Hi <= 0 || Wi <= 0) return MVSTER_ERR_SHAPE;
    if (prob_w && cout != 8) return MVSTER_ERR_UNSUPPORTED;
    const long in_elems = (long)NB * Hi * Wi * cin;
    if (in_elems >= (1L << 30) || (long)NB * Hi * Wi * 4 * cout >= (1L << 31)) return MVSTER_ERR_SHAPE;
    DeconvArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.skip = skip; a.prob_w = prob_w; a.prob_b = prob_b;
    a.out = out; a.NB = NB; a.Hi = Hi; a.Wi = Wi; a.relu = relu; a.in_bytes = (unsigned)(in_elems * 4);
    dim3 grid((Hi * Wi + 255) / 256, NB), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (cin == 16 && cout == 8) {
        MV_NOTE_KERNEL("deconv_small_kernel<16, 8>");
        hipLaunchKernelGGL((deconv_small_kernel<16, 8>), grid, block, 0, s, a);
    } else if (cin == 32 && cout == 16) {
        MV_NOTE_KERNEL("deconv_small_kernel<32, 16>");
        hipLaunchKernelGGL((deconv_small_kernel<32, 16>), grid, block, 0, s, a);
    } else {
        return MVSTER_ERR_UNSUPPORTED;
    }
    return mv_check_launch();
}

// mvster_deconv_small (16 -> 8, fused `prob` head) + mvster_select_depth in one launch: in [B*D,Hi,Wi,16] (slices b*D + d),
// skip [B*D,2Hi,2Wi,8] or null, hypo [B,D,2Hi,2Wi] -> attn [B,D,2Hi,2Wi], depth / conf / inv_min / inv_max [B,2Hi,2Wi]
// (conf, inv_* optional), logits_out [B,D,2Hi,2Wi] optional.  Bit-identical to the two launches.  D <= 16.
extern "C" int mvster_deconv_select(const float* in, const float* w, const float* scale, const float* shift, const float* skip,
                                    const float* prob_w, const float* prob_b, const float* hypo, float* attn, float* depth,
                                    float* conf, float* inv_min, float* inv_max, float* logits_out, int B, int D, int Hi,
                                    int Wi, int cin, int relu, float split_itv, void* stream) {
    if (!in || !w || !scale || !shift || !prob_w || !prob_b || !hypo || !attn || !depth) return MVSTER_ERR_NULL;
    if ((inv_min == nullptr) != (inv_max == nullptr)) return MVSTER_ERR_NULL;
    if (B <= 0 || D < 2 || D > mv::kSelMaxD || Hi <= 0 || Wi <= 0 || (inv_min && D < 3)) return MVSTER_ERR_SHAPE;
    if (cin != 16) return MVSTER_ERR_UNSUPPORTED;
    const long in_elems = (long)B * D * Hi * Wi * cin;
    if (in_elems >= (1L << 30) || (long)B * D * Hi * Wi * 4 * 8 >= (1L << 31)) return MVSTER_ERR_SHAPE;
    {
        // D in {4, 8} (the shipped cascade): MFMA tiles on the persistent LDS-DMA ring, same bits (deconv_select.hip)
        const int rc = mvconv::dispatch_deconv_select_mfma(in, w, scale, shift, skip, prob_w, prob_b, hypo, attn, depth, conf, inv_min,
                                                           inv_max, logits_out, B, D, Hi, Wi, relu, split_itv, (hipStream_t)stream);
        if (rc != MVSTER_ERR_UNSUPPORTED) return rc;
    }
    DeconvArgs a;
    a.in = in; a.w = w; a.scale = scale; a.shift = shift; a.skip = skip; a.prob_w = prob_w; a.prob_b = prob_b;
    a.out = nullptr; a.NB = B * D; a.Hi = Hi; a.Wi = Wi; a.relu = relu; a.in_bytes = (unsigned)(in_elems * 4);
    SelectOut so;
    so.hypo = hypo; so.attn = attn; so.depth = depth; so.conf = conf; so.inv_min = inv_min; so.inv_max = inv_max;
    so.logits_out = logits_out; so.D = D; so.split_itv = split_itv;
    MV_NOTE_KERNEL("deconv_select_kernel<16>");
    hipLaunchKernelGGL(deconv_select_kernel<16>, dim3((Hi * Wi + 63) / 64, B), dim3(64, D), 0, (hipStream_t)stream, a, so);
    return mv_check_launch();
}
