// Per-element arithmetic of the MVSTER cost-volume path, shared by every kernel.
//
// Everything here is a small inline function over scalars so that (a) the HIP
// kernels in this directory and (b) a host build used only by the tests
// (tests/hostmath) execute the *same* expression trees.  The op order mirrors the
// reference's PyTorch expressions so that fp32 results agree to rounding:
//   projection   models/mvs4net_utils.py:34-45   (rot*xyz, *depth, +trans, z==0 fix, /z, normalise)
//   sampling     F.grid_sample(bilinear, zeros, align_corners=True)  (mvs4net_utils.py:51)
//   correlation  models/mvs4net_utils.py:1038-1042
//   attention    models/mvs4net_utils.py:1053-1060
//   upsample     F.interpolate(..., align_corners=True)  (mvs4net_utils.py:85, :1077)
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MV_HD __host__ __device__ __forceinline__
#else
#define MV_HD inline
#endif

// Division by a launch-time constant as multiply-high + shift (exact for n < 2^31; tests/test_hostmath.py checks it
// against `/` on the host build).  A runtime `/` or `%` costs a wavefront ~20-40 instructions (reciprocal + fix-ups); the
// small kernels of this path do a handful per thread, and in the LDS convolution's prologue they were a third of all
// instructions.  The host fills the struct (mv_fastdiv), kernels take it by value.
struct FastDiv {
    unsigned d, mul, shr;
};

static inline FastDiv mv_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
    int lg = 31 - __builtin_clz(d);
    if (d & (d - 1)) ++lg;                       // ceil(log2 d)
    const int p = 31 + lg;
    f.mul = (unsigned)(((1ull << p) + d - 1) / d);
    f.shr = (unsigned)(p - 32);
    return f;
}

// the same constants computed in a kernel (divisors that only exist on the device: a tile's patch width)
MV_HD FastDiv mv_fastdiv_dev(unsigned d) {
    FastDiv f;
    f.d = d;
    f.mul = 0;
    f.shr = 0;
    if (d <= 1) return f;
    int lg = 0;
    while ((1u << lg) < d) ++lg;                 // ceil(log2 d)
    const int p = 31 + lg;
    f.mul = (unsigned)(((1ull << p) + d - 1) / d);
    f.shr = (unsigned)(p - 32);
    return f;
}

MV_HD unsigned fdiv(unsigned n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr);
#else
    return f.d == 1 ? n : ((unsigned)(((unsigned long long)n * f.mul) >> 32) >> f.shr);
#endif
}
// quotient and remainder
MV_HD unsigned fdivmod(unsigned n, const FastDiv& f, unsigned& rem) {
    const unsigned q = fdiv(n, f);
    rem = n - q * f.d;
    return q;
}

namespace mv {

// fp32 ops that must not be contracted into FMAs (the reference materialises the
// intermediate tensors, so every product and sum is rounded on its own).  hipcc's
// __fmul_rn/__fadd_rn are plain `*`/`+` and WOULD be contracted; the pragma removes the
// `contract` flag from the instruction itself, which survives inlining (hipcc's default is
// -ffp-contract=fast-honor-pragmas).  The host test build uses -ffp-contract=off.
MV_HD float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
MV_HD float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
MV_HD float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
MV_HD float div_rn(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}

// Division with a reusable reciprocal.  On the device this is the IEEE-754 fp32 division sequence
// of the AMDGPU backend (v_rcp_f32, one Newton step on the reciprocal, quotient, two residual
// corrections) WITHOUT its v_div_scale / v_div_fmas / v_div_fixup range handling: for normal,
// non-zero divisors and quotients that neither overflow nor go subnormal it returns the same bits
// as `a / b` (tests/test_gpu_kernels.py compares kernels built on both), costs 5 instructions per
// quotient instead of 11, and the 3-instruction reciprocal is shared by every quotient with the
// same divisor.  Callers guarantee the domain (depths, image sizes, softmax masses).  The host build
// used by the tests is plain division.
struct Recip {
    float d;   // divisor
    float r;   // refined reciprocal (device only)
};

MV_HD Recip make_recip(float d) {
    Recip k;
    k.d = d;
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(d);
    k.r = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
#else
    k.r = 0.0f;
#endif
    return k;
}

MV_HD float div_rn(float a, const Recip& k) {
#if defined(__HIP_DEVICE_COMPILE__)
    float q = mul_rn(a, k.r);
    q = __builtin_fmaf(__builtin_fmaf(-k.d, q, a), k.r, q);
    return __builtin_fmaf(__builtin_fmaf(-k.d, q, a), k.r, q);
#else
    return div_rn(a, k.d);
#endif
}

// 3x4 homography of one (batch, source view): p_src ~ R * (x, y, 1) * depth + t.
struct RT {
    float r[9];
    float t[3];
};

// Normalised->pixel round trip of the reference: the grid is normalised with
// (size-1)/2 (mvs4net_utils.py:43-44) and ATen un-normalises it with
// ((g + 1) / 2) * (size - 1) (GridSampler.h, align_corners=True).
MV_HD float grid_roundtrip(float pix, int size) {
    float half = (float)(size - 1) / 2.0f;          // exact in fp32 for any feature-map size
    float g = sub_rn(div_rn(pix, half), 1.0f);
    return mul_rn(div_rn(add_rn(g, 1.0f), 2.0f), (float)(size - 1));
}

// Source-view sampling position (in source pixels, after the round trip) of
// reference pixel (x, y) at hypothesis `depth`.
MV_HD void project(const RT& m, float x, float y, float depth, int Hs, int Ws, float& sx, float& sy) {
    // rot @ (x, y, 1) is a GEMM in the reference: an FMA chain over k (bit-identical to
    // torch's CPU sgemm on the golden vectors, tests/test_hostmath.py)
    float rx = add_rn(fmaf(m.r[1], y, mul_rn(m.r[0], x)), m.r[2]);
    float ry = add_rn(fmaf(m.r[4], y, mul_rn(m.r[3], x)), m.r[5]);
    float rz = add_rn(fmaf(m.r[7], y, mul_rn(m.r[6], x)), m.r[8]);
    float px = add_rn(mul_rn(rx, depth), m.t[0]);
    float py = add_rn(mul_rn(ry, depth), m.t[1]);
    float pz = add_rn(mul_rn(rz, depth), m.t[2]);
    if (pz == 0.0f) pz = 1e-9f;
    sx = grid_roundtrip(div_rn(px, pz), Ws);
    sy = grid_roundtrip(div_rn(py, pz), Hs);
}

// The same position with the per-launch constants of the round trip hoisted and shared reciprocals
// (kernels; bit-identical to project() in the domain described at Recip).
struct GridNorm {
    Recip halfw, halfh;   // (Ws-1)/2, (Hs-1)/2
    float wm1, hm1;       // Ws-1, Hs-1
};

MV_HD GridNorm make_grid_norm(int Hs, int Ws) {
    GridNorm g;
    g.halfw = make_recip((float)(Ws - 1) / 2.0f);
    g.halfh = make_recip((float)(Hs - 1) / 2.0f);
    g.wm1 = (float)(Ws - 1);
    g.hm1 = (float)(Hs - 1);
    return g;
}

MV_HD float grid_roundtrip(float pix, const Recip& half, float sizem1) {
    float g = sub_rn(div_rn(pix, half), 1.0f);
    return mul_rn(mul_rn(add_rn(g, 1.0f), 0.5f), sizem1);   // x / 2 == x * 0.5 exactly
}

MV_HD void project(const RT& m, float x, float y, float depth, const GridNorm& gn, float& sx, float& sy) {
    float rx = add_rn(fmaf(m.r[1], y, mul_rn(m.r[0], x)), m.r[2]);
    float ry = add_rn(fmaf(m.r[4], y, mul_rn(m.r[3], x)), m.r[5]);
    float rz = add_rn(fmaf(m.r[7], y, mul_rn(m.r[6], x)), m.r[8]);
    float px = add_rn(mul_rn(rx, depth), m.t[0]);
    float py = add_rn(mul_rn(ry, depth), m.t[1]);
    float pz = add_rn(mul_rn(rz, depth), m.t[2]);
    if (pz == 0.0f) pz = 1e-9f;
    const Recip z = make_recip(pz);
    sx = grid_roundtrip(div_rn(px, z), gn.halfw, gn.wm1);
    sy = grid_roundtrip(div_rn(py, z), gn.halfh, gn.hm1);
}

// Bilinear footprint: integer corner, the four weights, and which taps are in bounds
// (zeros padding is applied per tap, like ATen's within_bounds_2d).
struct Taps {
    int x0, y0;            // north-west corner
    float nw, ne, sw, se;  // weights
    bool vx0, vx1, vy0, vy1;
};

MV_HD Taps make_taps(float sx, float sy, int Hs, int Ws) {
    Taps t;
    // clamp far-away / non-finite positions so that the int conversion is defined; anything beyond
    // one pixel outside samples only zeros anyway (fminf/fmaxf drop a NaN operand: NaN -> size + 4)
    float cx = fmaxf(fminf(sx, (float)(Ws + 4)), -4.0f);
    float cy = fmaxf(fminf(sy, (float)(Hs + 4)), -4.0f);
    float fx = floorf(cx), fy = floorf(cy);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    float wx1 = sub_rn(cx, fx), wy1 = sub_rn(cy, fy);   // east / south
    float wx0 = sub_rn(1.0f, wx1), wy0 = sub_rn(1.0f, wy1);
    t.nw = mul_rn(wy0, wx0);
    t.ne = mul_rn(wy0, wx1);
    t.sw = mul_rn(wy1, wx0);
    t.se = mul_rn(wy1, wx1);
    t.vx0 = (unsigned)t.x0 < (unsigned)Ws;
    t.vx1 = (unsigned)(t.x0 + 1) < (unsigned)Ws;
    t.vy0 = (unsigned)t.y0 < (unsigned)Hs;
    t.vy1 = (unsigned)(t.y0 + 1) < (unsigned)Hs;
    return t;
}

// The east / south fractions make_taps() forms its weights from, and the weights rebuilt from them (same operations,
// same bits): what the sorted scatter of the warp backward stores per sample instead of four weights.
MV_HD void tap_fractions(float sx, float sy, int Hs, int Ws, float& wx1, float& wy1) {
    float cx = fmaxf(fminf(sx, (float)(Ws + 4)), -4.0f);
    float cy = fmaxf(fminf(sy, (float)(Hs + 4)), -4.0f);
    wx1 = sub_rn(cx, floorf(cx));
    wy1 = sub_rn(cy, floorf(cy));
}
MV_HD void tap_weights(float wx1, float wy1, float& nw, float& ne, float& sw, float& se) {
    float wx0 = sub_rn(1.0f, wx1), wy0 = sub_rn(1.0f, wy1);
    nw = mul_rn(wy0, wx0);
    ne = mul_rn(wy0, wx1);
    sw = mul_rn(wy1, wx0);
    se = mul_rn(wy1, wx1);
}

// Branch-free form for the kernels: out-of-bounds taps get weight 0 and a clamped (always
// readable) address instead of a predicated load.  0 * finite == 0, so the blend is the
// same value as with a zero-padded tap.
struct TapsClamped {
    int xa, xb, ya, yb;  // clamped west/east column, north/south row
};

MV_HD int clampi(int v, int hi) {   // clamp to [0, hi] (v_med3_i32)
    v = v < hi ? v : hi;
    return v > 0 ? v : 0;
}

MV_HD TapsClamped clamp_taps(Taps& t, int Hs, int Ws) {
    TapsClamped c;
    if (!(t.vy0 && t.vx0)) t.nw = 0.0f;
    if (!(t.vy0 && t.vx1)) t.ne = 0.0f;
    if (!(t.vy1 && t.vx0)) t.sw = 0.0f;
    if (!(t.vy1 && t.vx1)) t.se = 0.0f;
    c.xa = clampi(t.x0, Ws - 1);
    c.xb = clampi(t.x0 + 1, Ws - 1);
    c.ya = clampi(t.y0, Hs - 1);
    c.yb = clampi(t.y0 + 1, Hs - 1);
    return c;
}

// ((nw*a + ne*b) + sw*c) + se*d, each product rounded (ATen CPU/CUDA order).
MV_HD float blend(const Taps& t, float a, float b, float c, float d) {
    return add_rn(add_rn(add_rn(mul_rn(a, t.nw), mul_rn(b, t.ne)), mul_rn(c, t.sw)), mul_rn(d, t.se));
}

#if defined(__HIPCC__)
// Two adjacent channels at a time (v_pk_mul_f32 / v_pk_add_f32 on gfx950): the same per-element
// expression tree as blend(), on the register pairs a 16-byte load delivers.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 mul_rn2(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ f32x2 add_rn2(f32x2 a, f32x2 b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ f32x2 blend2(float wnw, float wne, float wsw, float wse, f32x2 a, f32x2 b, f32x2 c, f32x2 d) {
    const f32x2 nw = {wnw, wnw}, ne = {wne, wne}, sw = {wsw, wsw}, se = {wse, wse};
    return add_rn2(add_rn2(add_rn2(mul_rn2(a, nw), mul_rn2(b, ne)), mul_rn2(c, sw)), mul_rn2(d, se));
}
#endif

// 1-D linear upsampling coefficient, align_corners=True (ATen area_pixel_compute_source_index
// + guard_index_and_lambda): src = dst * (in-1)/(out-1).
struct Lerp {
    int i0, i1;
    float w0, w1;
};

MV_HD Lerp make_lerp(int dst, int in_size, int out_size) {
    Lerp l;
    float scale = (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
    float src = mul_rn(scale, (float)dst);
    int i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    float lam = sub_rn(src, (float)i0);
    lam = lam < 0.0f ? 0.0f : (lam > 1.0f ? 1.0f : lam);
    l.i0 = i0;
    l.i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l.w1 = lam;
    l.w0 = sub_rn(1.0f, lam);
    return l;
}

// make_lerp with the scale (in_size - 1) / (out_size - 1) formed once by the caller (same fp32 quotient: same bits)
MV_HD float lerp_scale(int in_size, int out_size) { return (out_size > 1) ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f; }
MV_HD Lerp make_lerp_s(int dst, float scale, int in_size) {
    Lerp l;
    float src = mul_rn(scale, (float)dst);
    int i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    float lam = sub_rn(src, (float)i0);
    lam = lam < 0.0f ? 0.0f : (lam > 1.0f ? 1.0f : lam);
    l.i0 = i0;
    l.i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    l.w1 = lam;
    l.w0 = sub_rn(1.0f, lam);
    return l;
}

// h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
MV_HD float bilerp(const Lerp& ly, const Lerp& lx, float v00, float v01, float v10, float v11) {
    float top = add_rn(mul_rn(lx.w0, v00), mul_rn(lx.w1, v01));
    float bot = add_rn(mul_rn(lx.w0, v10), mul_rn(lx.w1, v11));
    return add_rn(mul_rn(ly.w0, top), mul_rn(ly.w1, bot));
}

// double-precision 4x4 inverse (Gauss-Jordan, partial pivoting); returns false if singular
MV_HD bool inverse4(const double* a, double* inv) {
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            m[i][j] = a[i * 4 + j];
            m[i][j + 4] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = fabs(m[c][c]);
        for (int r = c + 1; r < 4; ++r)
            if (fabs(m[r][c]) > best) { best = fabs(m[r][c]); p = r; }
        if (best == 0.0) return false;
        if (p != c)
            for (int j = 0; j < 8; ++j) { double tmp = m[c][j]; m[c][j] = m[p][j]; m[p][j] = tmp; }
        double d = 1.0 / m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] *= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            double f = m[r][c];
            if (f != 0.0)
                for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i * 4 + j] = m[i][j + 4];
    return true;
}

// Compose K @ [R|t] for one camera: pm = {extrinsic 4x4, intrinsic 4x4} (fp32, row major).
// Rows 0..2 of the result are K[:3,:3] @ E[:3,:4] rounded to fp32 like the reference's fp32
// matmul (mvs4net_utils.py:1033), row 3 is the extrinsic's.
MV_HD void compose_camera(const float* pm, double* P) {
    const float* E = pm;
    const float* K = pm + 16;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 3; ++k) acc = fmaf(K[i * 4 + k], E[k * 4 + j], acc);
            P[i * 4 + j] = (double)acc;
        }
    for (int j = 0; j < 4; ++j) P[12 + j] = (double)E[12 + j];
}

// src_P @ inv(ref_P), top 3x4 rounded to fp32 (mvs4net_utils.py:24-26).  A singular reference projection
// (torch.inverse raises there) yields an all-NaN result, which every consumer propagates into its outputs:
// a degenerate camera is loud, never a silently wrong warp.
MV_HD bool relative_projection(const float* ref_pm, const float* src_pm, RT& out) {
    double Pr[16], Ps[16], Pi[16];
    compose_camera(ref_pm, Pr);
    compose_camera(src_pm, Ps);
    bool ok = inverse4(Pr, Pi);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += Ps[i * 4 + k] * Pi[k * 4 + j];
            if (!ok) acc = (double)NAN;
            if (j < 3) out.r[i * 3 + j] = (float)acc;
            else out.t[i] = (float)acc;
        }
    return ok;
}


// ---------------------------------------------------------------------------------------
// Per-pixel bodies of the small stage kernels (stage_ops.hip); `p` is the linear pixel index.
// ---------------------------------------------------------------------------------------

// init_inverse_range / init_range (mvs4net_utils.py:61-77): out[d*hw + p] for one batch item
// one hypothesis of init_inverse_range (the loop below and the warp kernel's fused form share it: same bits)
MV_HD float init_inverse_one(float dmin, float dmax, int D, int d) {
    const float inv_near = div_rn(1.0f, dmin), inv_far = div_rn(1.0f, dmax);
    const float span = sub_rn(inv_near, inv_far);
    const float itv = div_rn((float)d, (float)(D - 1));
    return div_rn(1.0f, add_rn(inv_far, mul_rn(span, itv)));
}

MV_HD void init_range_pixel(float dmin, float dmax, float* out, int D, long hw, long p, int inverse) {
    if (inverse) {
        for (int d = 0; d < D; ++d) out[d * hw + p] = init_inverse_one(dmin, dmax, D, d);
    } else {
        const float step = div_rn(sub_rn(dmax, dmin), (float)(D - 1));
        for (int d = 0; d < D; ++d) out[d * hw + p] = add_rn(dmin, mul_rn((float)d, step));
    }
}

// schedule_inverse_range (mvs4net_utils.py:79-86): inv_min/inv_max are one batch item's
// [hi, wi] maps; trilinear with the depth size unchanged is bilinear per slice.
// the four corners of a pixel's inverse-depth range at the previous stage's resolution
struct InvCorners { Lerp ly, lx; float mx[4], sp[4]; };
MV_HD InvCorners schedule_inverse_corners(const float* inv_min, const float* inv_max, int y, int x, int h, int w, int hi, int wi) {
    InvCorners c;
    c.ly = make_lerp(y, hi, h);
    c.lx = make_lerp(x, wi, w);
    const int idx[4] = {c.ly.i0 * wi + c.lx.i0, c.ly.i0 * wi + c.lx.i1, c.ly.i1 * wi + c.lx.i0, c.ly.i1 * wi + c.lx.i1};
    for (int k = 0; k < 4; ++k) {
        c.mx[k] = inv_max[idx[k]];
        c.sp[k] = sub_rn(inv_min[idx[k]], c.mx[k]);
    }
    return c;
}
// one hypothesis of schedule_inverse_range (the loop below and the warp kernel's fused form share it: same bits)
MV_HD float schedule_inverse_one(const InvCorners& c, int D, int d) {
    const float itv = div_rn((float)d, (float)(D - 1));
    const float v00 = add_rn(c.mx[0], mul_rn(c.sp[0], itv)), v01 = add_rn(c.mx[1], mul_rn(c.sp[1], itv));
    const float v10 = add_rn(c.mx[2], mul_rn(c.sp[2], itv)), v11 = add_rn(c.mx[3], mul_rn(c.sp[3], itv));
    return div_rn(1.0f, bilerp(c.ly, c.lx, v00, v01, v10, v11));
}

MV_HD void schedule_inverse_pixel(const float* inv_min, const float* inv_max, float* out, int D, int h, int w,
                                  int hi, int wi, int p) {
    const int y = p / w, x = p - y * w;
    const InvCorners c = schedule_inverse_corners(inv_min, inv_max, y, x, h, w, hi, wi);
    for (int d = 0; d < D; ++d) out[(long)d * h * w + p] = schedule_inverse_one(c, D, d);
}

// schedule_range (mvs4net_utils.py:88-99)
MV_HD void schedule_linear_pixel(const float* cur, float interval, float* out, int D, int h, int w, int hi, int wi,
                                 int p) {
    const int y = p / w, x = p - y * w;
    const Lerp ly = make_lerp(y, hi, h), lx = make_lerp(x, wi, w);
    const int idx[4] = {ly.i0 * wi + lx.i0, ly.i0 * wi + lx.i1, ly.i1 * wi + lx.i0, ly.i1 * wi + lx.i1};
    const float half = mul_rn((float)D / 2.0f, interval);  // ndepth / 2 * interval
    float lo[4], st[4];
    for (int k = 0; k < 4; ++k) {
        const float c = cur[idx[k]];
        lo[k] = sub_rn(c, half);
        st[k] = div_rn(sub_rn(add_rn(c, half), lo[k]), (float)(D - 1));
    }
    for (int d = 0; d < D; ++d) {
        float v[4];
        for (int k = 0; k < 4; ++k) v[k] = add_rn(lo[k], mul_rn((float)d, st[k]));
        out[(long)d * h * w + p] = bilerp(ly, lx, v[0], v[1], v[2], v[3]);
    }
}

constexpr int kSelMaxD = 16;

// prob 1x1x1 (optional) + softmax over D + first-max argmax + gather + inverse bounds
// (mvs4net_utils.py:900, :1068-1088).  All pointers are one batch item's; plane stride = hw.
// feat: [D, hw, CF] channels-last or null (then logits [D, hw] is read).
// softmax over D + first-max argmax + gather + inverse bounds from the D logits of one pixel (lg is overwritten)
// (hv = the pixel's D hypotheses, already in registers: the persistent kernel fetches them ahead of its MFMA phase)
MV_HD void select_from_logits_vals(float (&lg)[kSelMaxD], const float (&hv)[kSelMaxD], float* attn, float* depth, float* conf,
                                   float* inv_min, float* inv_max, int D, long hw, long p, float split_itv) {
    float mx = -INFINITY;
#pragma unroll
    for (int d = 0; d < kSelMaxD; ++d) {
        if (d >= D) break;
        mx = fmaxf(mx, lg[d]);
    }
    float den = 0.0f;
#pragma unroll
    for (int d = 0; d < kSelMaxD; ++d) {
        if (d >= D) break;
        lg[d] = expf(sub_rn(lg[d], mx));
        den = add_rn(den, lg[d]);
    }
    float best = -1.0f, hb = 0.f, h1 = 0.f, h2 = 0.f;
#pragma unroll
    for (int d = 0; d < kSelMaxD; ++d) {
        if (d >= D) break;
        const long o = d * hw + p;
        const float pr = div_rn(lg[d], den);
        attn[o] = pr;
        const float hd = hv[d];
        if (d == 1) h1 = hd;
        if (d == 2) h2 = hd;
        if (pr > best) { best = pr; hb = hd; }  // strict '>' : the first maximum wins ties (ATen max)
    }
    depth[p] = hb;
    if (conf) conf[p] = best;
    if (inv_min) {
        const float itv = sub_rn(div_rn(1.0f, h2), div_rn(1.0f, h1));
        const float inv_d = div_rn(1.0f, hb);
        const float delta = mul_rn(split_itv, itv);
        inv_min[p] = add_rn(inv_d, delta);
        inv_max[p] = sub_rn(inv_d, delta);
    }
}

MV_HD void select_from_logits(float (&lg)[kSelMaxD], const float* hypo, float* attn, float* depth, float* conf,
                              float* inv_min, float* inv_max, int D, long hw, long p, float split_itv) {
    float hv[kSelMaxD];
#pragma unroll
    for (int d = 0; d < kSelMaxD; ++d) {
        if (d >= D) break;
        hv[d] = hypo[d * hw + p];
    }
    select_from_logits_vals(lg, hv, attn, depth, conf, inv_min, inv_max, D, hw, p, split_itv);
}

MV_HD void select_pixel(const float* logits, const float* feat, const float* prob_w, const float* prob_b, int CF,
                        const float* hypo, float* attn, float* depth, float* conf, float* inv_min, float* inv_max,
                        float* logits_out, int D, long hw, long p, float split_itv) {
    float lg[kSelMaxD];  // fully unrolled + guarded so that it stays in registers on the GPU
#pragma unroll
    for (int d = 0; d < kSelMaxD; ++d) {
        if (d >= D) break;
        const long o = d * hw + p;
        float v;
        if (feat) {
            const float* f = feat + o * CF;
            v = 0.0f;
            for (int c = 0; c < CF; ++c) v = fmaf(f[c], prob_w[c], v);
            v = add_rn(v, prob_b[0]);
            if (logits_out) logits_out[o] = v;
        } else {
            v = logits[o];
        }
        lg[d] = v;
    }
    select_from_logits(lg, hypo, attn, depth, conf, inv_min, inv_max, D, hw, p, split_itv);
}

// The same selection for any number of hypotheses (D > kSelMaxD: a free --ndepths of the reference): three passes over the
// pixel's D logits through the attn plane instead of a register array; every value goes through the same operations in the
// same order as select_pixel, so the two agree bit for bit where both apply.
MV_HD void select_pixel_any(const float* logits, const float* feat, const float* prob_w, const float* prob_b, int CF,
                            const float* hypo, float* attn, float* depth, float* conf, float* inv_min, float* inv_max,
                            float* logits_out, int D, long hw, long p, float split_itv) {
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) {
        const long o = d * hw + p;
        float v;
        if (feat) {
            const float* f = feat + o * CF;
            v = 0.0f;
            for (int c = 0; c < CF; ++c) v = fmaf(f[c], prob_w[c], v);
            v = add_rn(v, prob_b[0]);
            if (logits_out) logits_out[o] = v;
        } else {
            v = logits[o];
        }
        attn[o] = v;
        mx = fmaxf(mx, v);
    }
    float den = 0.0f;
    for (int d = 0; d < D; ++d) {
        const long o = d * hw + p;
        const float e = expf(sub_rn(attn[o], mx));
        attn[o] = e;
        den = add_rn(den, e);
    }
    float best = -1.0f, hb = 0.f, h1 = 0.f, h2 = 0.f;
    for (int d = 0; d < D; ++d) {
        const long o = d * hw + p;
        const float pr = div_rn(attn[o], den);
        attn[o] = pr;
        const float hd = hypo[o];
        if (d == 1) h1 = hd;
        if (d == 2) h2 = hd;
        if (pr > best) { best = pr; hb = hd; }  // strict '>' : the first maximum wins ties (ATen max)
    }
    depth[p] = hb;
    if (conf) conf[p] = best;
    if (inv_min) {
        const float itv = sub_rn(div_rn(1.0f, h2), div_rn(1.0f, h1));
        const float inv_d = div_rn(1.0f, hb);
        const float delta = mul_rn(split_itv, itv);
        inv_min[p] = add_rn(inv_d, delta);
        inv_max[p] = sub_rn(inv_d, delta);
    }
}

// F.interpolate(bilinear, align_corners=True) of one [hi, wi] map at output pixel p
MV_HD float upsample_pixel(const float* in, int hi, int wi, int ho, int wo, int p) {
    const int y = p / wo, x = p - y * wo;
    const Lerp ly = make_lerp(y, hi, ho), lx = make_lerp(x, wi, wo);
    return bilerp(ly, lx, in[ly.i0 * wi + lx.i0], in[ly.i0 * wi + lx.i1], in[ly.i1 * wi + lx.i0],
                  in[ly.i1 * wi + lx.i1]);
}

}  // namespace mv
