// Fused Sinkhorn optimal-transport loss of the training objective.
//
// The reference (models/mvs4net_utils.py:1096-1142) materialises the [B, HW, D, D] cost and runs `iters`
// log-domain Sinkhorn updates as ~8 tensor passes each, all kept alive for autograd: ~0.6 GB of
// intermediates per stage-4 call and dozens of launches.  D <= 8 (shipped), so one thread can hold the whole problem
// of one pixel in registers: both potentials, the |i-j| cost analytically, the transport plan, and the
// reverse sweep through the iterations.  One launch returns the per-pixel loss AND its gradient with
// respect to the predicted distribution (the only differentiable input), so backward is a multiply.
//   a_j = log(onehot(nearest hypothesis to gt)_j + 1e-12)        b_i = log(pred_i + 1e-12)
//   v_j = a_j - LSE_i(K_ij + u_i),  u_i = b_i - LSE_j(K_ij + v_j),  K = |i-j| / eps     (iters times, u0 = 0)
//   loss = sum_ij exp(K_ij + u_i + v_j) * |i-j|
// The continuous variant (ot_continous, :1111-1123) has D + 1 target columns; see the kernel.
#include "common.hpp"

namespace {

constexpr int kMaxIters = 16;

// CONT = false: the discrete form (D target bins, one-hot on the hypothesis nearest to the ground truth).
// CONT = true: the continuous form (ot_continous, models/mvs4net_utils.py:1111-1123): E = D + 1 target columns, all the
// mass on the extra one, whose cost column is |pos - i| with pos = (1/gt - 1/hypo_0) / (1/hypo_2 - 1/hypo_1) the
// ground truth's fractional bin (10 where the pixel is masked out, like the reference; `mask` is only read then).
template <int D, bool CONT>
__global__ void __launch_bounds__(128) sinkhorn_kernel(const float* __restrict__ attn, const float* __restrict__ hypo,
                                                       const float* __restrict__ gt, const float* __restrict__ mask,
                                                       float* __restrict__ loss_pix, float* __restrict__ jac, int B, long HW,
                                                       int iters, float inv_eps) {
    constexpr int E = CONT ? D + 1 : D;
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)B * HW) return;
    const long b = p / HW, q = p - b * HW;
    const float* ap = attn + b * D * HW + q;
    const float* hp = hypo + b * D * HW + q;
    const float g = gt[p];
    float pred[D], bl[D], a[E], last[D];
    int nearest = 0;
    float best = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        pred[i] = ap[i * HW];
        bl[i] = logf(pred[i] + 1e-12f);
        const float dist = fabsf(hp[i * HW] - g);
        if (i == 0 || dist < best) { best = dist; nearest = i; }      // first minimum, like torch.min
    }
    if (CONT) {
        const float itv = 1.0f / hp[2 * HW] - 1.0f / hp[HW];
        float pos = (1.0f / g - 1.0f / hp[0]) / itv;
        if (!(mask[p] > 0.5f)) pos = 10.0f;
#pragma unroll
        for (int i = 0; i < D; ++i) last[i] = fabsf(pos - (float)i);
        nearest = D;
    }
#pragma unroll
    for (int j = 0; j < E; ++j) a[j] = logf((j == nearest ? 1.0f : 0.0f) + 1e-12f);
    // cost of moving bin i to target column j
    auto cost = [&](int i, int j) -> float { return (CONT && j == D) ? last[i] : fabsf((float)(i - j)); };

    float uh[kMaxIters][D], vh[kMaxIters][E];      // potentials after every iteration (reverse sweep)
    float u[D], v[E];
#pragma unroll
    for (int i = 0; i < D; ++i) u[i] = 0.0f;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < D; ++i) m = fmaxf(m, cost(i, j) * inv_eps + u[i]);
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) s += expf(cost(i, j) * inv_eps + u[i] - m);
            v[j] = a[j] - (m + logf(s));
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < E; ++j) m = fmaxf(m, cost(i, j) * inv_eps + v[j]);
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < E; ++j) s += expf(cost(i, j) * inv_eps + v[j] - m);
            u[i] = bl[i] - (m + logf(s));
        }
#pragma unroll
        for (int i = 0; i < D; ++i) uh[t][i] = u[i];
#pragma unroll
        for (int j = 0; j < E; ++j) vh[t][j] = v[j];
    }
    if (iters == 0) {
#pragma unroll
        for (int j = 0; j < E; ++j) v[j] = 0.0f;
    }
    // loss and the gradients of the final plan
    float loss = 0.0f, du[D], dv[E], db[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { du[i] = 0.0f; db[i] = 0.0f; }
#pragma unroll
    for (int j = 0; j < E; ++j) dv[j] = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const float c = cost(i, j);
            const float pc = expf(c * inv_eps + u[i] + v[j]) * c;
            loss += pc;
            du[i] += pc;
            dv[j] += pc;
        }
    loss_pix[p] = loss;
    // reverse sweep
    for (int t = iters - 1; t >= 0; --t) {
        float ut[D], vt[E], up[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ut[i] = uh[t][i];
            up[i] = t > 0 ? uh[t - 1][i] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < E; ++j) vt[j] = vh[t][j];
        // u_i = b_i - LSE_j(K_ij + v_j):  softmax_ij = exp(K_ij + v_j + u_i - b_i)
#pragma unroll
        for (int i = 0; i < D; ++i) db[i] += du[i];
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) acc += du[i] * expf(cost(i, j) * inv_eps + vt[j] + ut[i] - bl[i]);
            dv[j] -= acc;
        }
        // v_j = a_j - LSE_i(K_ij + u'_i) with u' the previous iterate:  softmax_ij = exp(K_ij + u'_i + v_j - a_j)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < E; ++j) acc += dv[j] * expf(cost(i, j) * inv_eps + up[i] + vt[j] - a[j]);
            du[i] = -acc;
        }
#pragma unroll
        for (int j = 0; j < E; ++j) dv[j] = 0.0f;
    }
    float* jp = jac + b * D * HW + q;
#pragma unroll
    for (int i = 0; i < D; ++i) jp[i * HW] = db[i] / (pred[i] + 1e-12f);
}

// The discrete form with the constant cost factored out of the exponentials: K_ij = |i - j| / eps does not depend on the
// pixel, so  LSE_i(K_ij + u_i) = m + log sum_i E_ij exp(u_i - m)  with E_k = exp(k / eps) (D constants) and m = max_i u_i
// (any shift is exact in exact arithmetic; 1 <= the sum <= D exp((D-1)/eps)).  D exponentials per update instead of D^2;
// the reverse sweep uses the same factorisation for its softmax matrices (exp(K_ij + v_j + u_i - b_i) is
// E_ij exp(v_j - m) / sum_j' E_ij' exp(v_j' - m) because u_i = b_i - LSE_j(K_ij + v_j)).  16 instead of 96 exponentials per
// iteration at D = 4, 32 instead of 384 at D = 8; same mathematics, results within rounding of the form above.
template <int D>
__global__ void __launch_bounds__(128) sinkhorn_fast_kernel(const float* __restrict__ attn, const float* __restrict__ hypo,
                                                            const float* __restrict__ gt, float* __restrict__ loss_pix,
                                                            float* __restrict__ jac, int B, long HW, int iters, float inv_eps) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)B * HW) return;
    const long b = p / HW, q = p - b * HW;
    const float* ap = attn + b * D * HW + q;
    const float* hp = hypo + b * D * HW + q;
    const float g = gt[p];
    float pred[D], bl[D], a[D], Ek[D];
    int nearest = 0;
    float best = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        pred[i] = ap[i * HW];
        bl[i] = logf(pred[i] + 1e-12f);
        const float dist = fabsf(hp[i * HW] - g);
        if (i == 0 || dist < best) { best = dist; nearest = i; }      // first minimum, like torch.min
        Ek[i] = expf((float)i * inv_eps);
    }
#pragma unroll
    for (int j = 0; j < D; ++j) a[j] = logf((j == nearest ? 1.0f : 0.0f) + 1e-12f);
#define MV_E(i, j) Ek[(i) > (j) ? (i) - (j) : (j) - (i)]
    float uh[kMaxIters][D], vh[kMaxIters][D];      // potentials after every iteration (reverse sweep)
    float u[D], v[D];
#pragma unroll
    for (int i = 0; i < D; ++i) u[i] = 0.0f;
    for (int t = 0; t < iters; ++t) {
        float m = u[0], e[D];
#pragma unroll
        for (int i = 1; i < D; ++i) m = fmaxf(m, u[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) e[i] = __expf(u[i] - m);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) sum = fmaf(MV_E(i, j), e[i], sum);
            v[j] = a[j] - (m + __logf(sum));
        }
        m = v[0];
#pragma unroll
        for (int j = 1; j < D; ++j) m = fmaxf(m, v[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) e[j] = __expf(v[j] - m);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) sum = fmaf(MV_E(i, j), e[j], sum);
            u[i] = bl[i] - (m + __logf(sum));
        }
#pragma unroll
        for (int i = 0; i < D; ++i) { uh[t][i] = u[i]; vh[t][i] = v[i]; }
    }
    if (iters == 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) v[j] = 0.0f;
    }
    // loss and the gradients of the final plan (D^2 exponentials, once)
    float loss = 0.0f, du[D], dv[D], db[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { du[i] = 0.0f; db[i] = 0.0f; dv[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float c = fabsf((float)(i - j));
            const float pc = expf(c * inv_eps + u[i] + v[j]) * c;
            loss += pc;
            du[i] += pc;
            dv[j] += pc;
        }
    loss_pix[p] = loss;
    // reverse sweep
    for (int t = iters - 1; t >= 0; --t) {
        float e[D], den[D];
        // u_i = b_i - LSE_j(K_ij + v_j): softmax_ij = E_ij exp(v_j - m) / sum_j' E_ij' exp(v_j' - m)
        float m = vh[t][0];
#pragma unroll
        for (int j = 1; j < D; ++j) m = fmaxf(m, vh[t][j]);
#pragma unroll
        for (int j = 0; j < D; ++j) e[j] = __expf(vh[t][j] - m);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) sum = fmaf(MV_E(i, j), e[j], sum);
            den[i] = du[i] / sum;
            db[i] += du[i];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) acc = fmaf(den[i], MV_E(i, j), acc);
            dv[j] -= acc * e[j];
        }
        // v_j = a_j - LSE_i(K_ij + u'_i), u' the previous iterate: softmax_ij = E_ij exp(u'_i - m) / sum_i' E_i'j exp(u'_i' - m)
        float up[D];
#pragma unroll
        for (int i = 0; i < D; ++i) up[i] = t > 0 ? uh[t - 1][i] : 0.0f;
        m = up[0];
#pragma unroll
        for (int i = 1; i < D; ++i) m = fmaxf(m, up[i]);
#pragma unroll
        for (int i = 0; i < D; ++i) e[i] = __expf(up[i] - m);
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) sum = fmaf(MV_E(i, j), e[i], sum);
            den[j] = dv[j] / sum;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) acc = fmaf(den[j], MV_E(i, j), acc);
            du[i] = -acc * e[i];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) dv[j] = 0.0f;
    }
#undef MV_E
    float* jp = jac + b * D * HW + q;
#pragma unroll
    for (int i = 0; i < D; ++i) jp[i * HW] = db[i] / (pred[i] + 1e-12f);
}

template <bool CONT>
int launch_sinkhorn(const float* attn, const float* hypo, const float* gt, const float* mask, float* loss_pix, float* jac,
                    int B, int D, long HW, int iters, float eps, hipStream_t s) {
    const long n = (long)B * HW;
    dim3 grid((unsigned)((n + 127) / 128)), block(128);
    const float inv_eps = 1.0f / eps;
    if constexpr (!CONT) {
        // the shipped stage widths on the factored form (D exponentials per update); others on the general kernel
#define MV_F(D_) if (D == D_) { hipLaunchKernelGGL((sinkhorn_fast_kernel<D_>), grid, block, 0, s, attn, hypo, gt, loss_pix, jac, B, HW, iters, inv_eps); return mv_check_launch(); }
        MV_F(4) MV_F(8)
#undef MV_F
    }
#define MV_S(D_) if (D == D_) { hipLaunchKernelGGL((sinkhorn_kernel<D_, CONT>), grid, block, 0, s, attn, hypo, gt, mask, loss_pix, jac, B, HW, iters, inv_eps); return mv_check_launch(); }
    if (!CONT) { MV_S(2) }
    MV_S(3) MV_S(4) MV_S(5) MV_S(6) MV_S(7) MV_S(8)
    // 9..16 hypotheses (MVS4net accepts stage_splits up to 16): same code; the potentials' history no longer fits the
    // register file and lives in scratch memory, so these instances are slower per pixel -- not a shipped configuration
    MV_S(9) MV_S(10) MV_S(11) MV_S(12) MV_S(13) MV_S(14) MV_S(15) MV_S(16)
#undef MV_S
    return MVSTER_ERR_UNSUPPORTED;
}

// The per-pixel terms of one stage of MVS4net_loss (models/MVS4Net.py:131-151) around the OT term, in one pass: the
// reference forms them with ~25 tensor ops per stage (mask compare, boolean gathers, reciprocals, |.|, <=, sum over D,
// ...), each a launch of a few microseconds in a training step.  terms [5][B*HW], valid = mask > 0.5:
//   0: valid ? 1 : 0                                   (sum = number of valid pixels)
//   1: valid ? |mono - gt| : 0                         (F.l1_loss(mono_depth[mask], depth_gt[mask]), :136-137)
//   2: valid ? (no hypothesis within one interval of gt) : 0       (mask_out_of_range, :141-147)
//   3: valid ? loss_pix : 0                            (the masked mean of the per-pixel OT loss)
//   4: valid ? sign(mono - gt) : 0                     (d |mono - gt| / d mono, for the backward pass)
// The interval is |t(hypo_2) - t(hypo_1)| with t = 1/x for inverse-depth ranges, x otherwise; a pixel is in range if
// |t(hypo_d) - t(gt)| <= interval for some d (comparisons with NaN are false, as in the reference).  mono may be null
// (planes 1 and 4 are then zero).  The caller reduces the planes with one sum.
__global__ void __launch_bounds__(256) stage_loss_terms_kernel(const float* __restrict__ hypo, const float* __restrict__ gt,
                                                               const float* __restrict__ mask,
                                                               const float* __restrict__ loss_pix,
                                                               const float* __restrict__ mono, float* __restrict__ terms,
                                                               int B, int D, long HW, int inverse) {
    const long n = (long)B * HW;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const long b = p / HW, q = p - b * HW;
    const float* hp = hypo + b * D * HW + q;
    const bool valid = mask[p] > 0.5f;
    const float g = gt[p];
    const float tg = inverse ? 1.0f / g : g;
    const float t1 = inverse ? 1.0f / hp[HW] : hp[HW], t2 = inverse ? 1.0f / hp[2 * HW] : hp[2 * HW];
    const float itv = fabsf(t2 - t1);
    bool inside = false;
    for (int d = 0; d < D; ++d) {
        const float h = hp[d * HW];
        const float t = inverse ? 1.0f / h : h;
        inside = inside || (fabsf(t - tg) <= itv);
    }
    float l1 = 0.0f, sg = 0.0f;
    if (mono) {
        const float diff = mono[p] - g;
        l1 = fabsf(diff);
        sg = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
    }
    terms[p] = valid ? 1.0f : 0.0f;
    terms[n + p] = valid ? l1 : 0.0f;
    terms[2 * n + p] = (valid && !inside) ? 1.0f : 0.0f;
    terms[3 * n + p] = valid ? loss_pix[p] : 0.0f;
    terms[4 * n + p] = valid ? sg : 0.0f;
}

}  // namespace

// attn, hypo [B,D,HW]; gt [B,HW] -> loss_pix [B,HW], jac [B,D,HW] = d loss_pix / d attn.  D in {2,...,16}, iters <= 16.
extern "C" int mvster_sinkhorn(const float* attn, const float* hypo, const float* gt, float* loss_pix, float* jac, int B,
                               int D, long HW, int iters, float eps, void* stream) {
    if (!attn || !hypo || !gt || !loss_pix || !jac) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0 || iters < 0 || !(eps > 0.0f)) return MVSTER_ERR_SHAPE;
    if (iters > kMaxIters) return MVSTER_ERR_UNSUPPORTED;
    return launch_sinkhorn<false>(attn, hypo, gt, nullptr, loss_pix, jac, B, D, HW, iters, eps, (hipStream_t)stream);
}

// The continuous form (ot_continous=True): as above with mask [B,HW] (> 0.5 = valid); D in {3,...,16}.
extern "C" int mvster_sinkhorn_continuous(const float* attn, const float* hypo, const float* gt, const float* mask,
                                          float* loss_pix, float* jac, int B, int D, long HW, int iters, float eps,
                                          void* stream) {
    if (!attn || !hypo || !gt || !mask || !loss_pix || !jac) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0 || iters < 0 || !(eps > 0.0f)) return MVSTER_ERR_SHAPE;
    if (iters > kMaxIters) return MVSTER_ERR_UNSUPPORTED;
    return launch_sinkhorn<true>(attn, hypo, gt, mask, loss_pix, jac, B, D, HW, iters, eps, (hipStream_t)stream);
}

// hypo [B,D,HW] (D >= 3), gt, mask, loss_pix [B,HW], mono [B,HW] or null -> terms [5][B*HW] (see the kernel).
extern "C" int mvster_stage_loss_terms(const float* hypo, const float* gt, const float* mask, const float* loss_pix,
                                       const float* mono, float* terms, int B, int D, long HW, int inverse_depth,
                                       void* stream) {
    if (!hypo || !gt || !mask || !loss_pix || !terms) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0 || D < 3) return MVSTER_ERR_SHAPE;
    const long n = (long)B * HW;
    if ((n + 255) / 256 >= (1L << 31)) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(stage_loss_terms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, hypo, gt,
                       mask, loss_pix, mono, terms, B, D, HW, inverse_depth ? 1 : 0);
    return mv_check_launch();
}
