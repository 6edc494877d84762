// Host build of mvster_amd/csrc/mvster_math.h -- TEST INFRASTRUCTURE ONLY.
// Lets the CPU test-suite check the per-element arithmetic the HIP kernels use
// against the golden vectors without a GPU.  Never loaded by the product.
#include "../../mvster_amd/csrc/mvster_math.h"

extern "C" {

// quotient of the multiply-shift division the kernels use (mvster_math.h FastDiv)
unsigned hm_fastdiv(unsigned n, unsigned d) {
    const FastDiv f = mv_fastdiv(d);
    return fdiv(n, f);
}
// number of n in [lo, hi) (stepping by `step`) whose quotient / remainder differ from the plain operators
long hm_fastdiv_mismatches(unsigned d, unsigned lo, unsigned hi, unsigned step) {
    const FastDiv f = mv_fastdiv(d);
    long bad = 0;
    for (unsigned long long n = lo; n < hi; n += step) {
        unsigned rem;
        const unsigned q = fdivmod((unsigned)n, f, rem);
        if (q != (unsigned)n / d || rem != (unsigned)n % d) ++bad;
    }
    return bad;
}


// ref_pm/src_pm: [2,4,4] each; out: 12 floats (r[9], t[3])
int hm_relative_projection(const float* ref_pm, const float* src_pm, float* out) {
    mv::RT m;
    bool ok = mv::relative_projection(ref_pm, src_pm, m);
    for (int i = 0; i < 9; ++i) out[i] = m.r[i];
    for (int i = 0; i < 3; ++i) out[9 + i] = m.t[i];
    return ok ? 0 : -1;
}

// fea [C,Hs,Ws], rt 12 floats, depth [D,Hr,Wr] -> out [C,D,Hr,Wr]
void hm_warp(const float* fea, const float* rt, const float* depth, float* out, int C, int D, int Hr, int Wr,
             int Hs, int Ws) {
    mv::RT m;
    for (int i = 0; i < 9; ++i) m.r[i] = rt[i];
    for (int i = 0; i < 3; ++i) m.t[i] = rt[9 + i];
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < Hr; ++y)
            for (int x = 0; x < Wr; ++x) {
                float sx, sy;
                mv::project(m, (float)x, (float)y, depth[(d * Hr + y) * Wr + x], Hs, Ws, sx, sy);
                mv::Taps t = mv::make_taps(sx, sy, Hs, Ws);
                for (int c = 0; c < C; ++c) {
                    const float* p = fea + (long)c * Hs * Ws;
                    float a = (t.vy0 && t.vx0) ? p[t.y0 * Ws + t.x0] : 0.f;
                    float b = (t.vy0 && t.vx1) ? p[t.y0 * Ws + t.x0 + 1] : 0.f;
                    float cc = (t.vy1 && t.vx0) ? p[(t.y0 + 1) * Ws + t.x0] : 0.f;
                    float dd = (t.vy1 && t.vx1) ? p[(t.y0 + 1) * Ws + t.x0 + 1] : 0.f;
                    out[(((long)c * D + d) * Hr + y) * Wr + x] = mv::blend(t, a, b, cc, dd);
                }
            }
}

// in [Hi,Wi] -> out [Ho,Wo], bilinear align_corners=True
void hm_upsample(const float* in, float* out, int Hi, int Wi, int Ho, int Wo) {
    for (int p = 0; p < Ho * Wo; ++p) out[p] = mv::upsample_pixel(in, Hi, Wi, Ho, Wo, p);
}

void hm_init_range(float dmin, float dmax, float* out, int D, int hw, int inverse) {
    for (int p = 0; p < hw; ++p) mv::init_range_pixel(dmin, dmax, out, D, hw, p, inverse);
}

void hm_schedule_inverse(const float* inv_min, const float* inv_max, float* out, int D, int h, int w) {
    for (int p = 0; p < h * w; ++p) mv::schedule_inverse_pixel(inv_min, inv_max, out, D, h, w, h / 2, w / 2, p);
}

void hm_schedule_linear(const float* cur, float interval, float* out, int D, int h, int w) {
    for (int p = 0; p < h * w; ++p) mv::schedule_linear_pixel(cur, interval, out, D, h, w, h / 2, w / 2, p);
}

void hm_select(const float* logits, const float* feat, const float* prob_w, const float* prob_b, int CF,
               const float* hypo, float* attn, float* depth, float* conf, float* inv_min, float* inv_max,
               float* logits_out, int D, int hw, float split_itv) {
    for (int p = 0; p < hw; ++p)
        mv::select_pixel(logits, feat, prob_w, prob_b, CF, hypo, attn, depth, conf, inv_min, inv_max, logits_out, D, hw,
                         p, split_itv);
}

void hm_select_any(const float* logits, const float* feat, const float* prob_w, const float* prob_b, int CF,
                   const float* hypo, float* attn, float* depth, float* conf, float* inv_min, float* inv_max,
                   float* logits_out, int D, int hw, float split_itv) {
    for (int p = 0; p < hw; ++p)
        mv::select_pixel_any(logits, feat, prob_w, prob_b, CF, hypo, attn, depth, conf, inv_min, inv_max, logits_out, D, hw,
                             p, split_itv);
}
}
