// Fused Sinkhorn optimal-transport loss of the training objective.
//
// The reference (models/mvs4net_utils.py:1096-1142) materialises the [B, HW, D, D] cost and runs `iters`
// log-domain Sinkhorn updates as ~8 tensor passes each, all kept alive for autograd: ~0.6 GB of
// intermediates per stage-4 call and dozens of launches.  D <= 8, so one thread can hold the whole problem
// of one pixel in registers: both potentials, the |i-j| cost analytically, the transport plan, and the
// reverse sweep through the iterations.  One launch returns the per-pixel loss AND its gradient with
// respect to the predicted distribution (the only differentiable input), so backward is a multiply.
//   a_j = log(onehot(nearest hypothesis to gt)_j + 1e-12)        b_i = log(pred_i + 1e-12)
//   v_j = a_j - LSE_i(K_ij + u_i),  u_i = b_i - LSE_j(K_ij + v_j),  K = |i-j| / eps     (iters times, u0 = 0)
//   loss = sum_ij exp(K_ij + u_i + v_j) * |i-j|
#include "common.hpp"

namespace {

constexpr int kMaxIters = 16;

template <int D>
__global__ void __launch_bounds__(128) sinkhorn_kernel(const float* __restrict__ attn, const float* __restrict__ hypo,
                                                       const float* __restrict__ gt, float* __restrict__ loss_pix,
                                                       float* __restrict__ jac, int B, long HW, int iters, float inv_eps) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)B * HW) return;
    const long b = p / HW, q = p - b * HW;
    const float* ap = attn + b * D * HW + q;
    const float* hp = hypo + b * D * HW + q;
    const float g = gt[p];
    float pred[D], bl[D], a[D];
    int nearest = 0;
    float best = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        pred[i] = ap[i * HW];
        bl[i] = logf(pred[i] + 1e-12f);
        const float dist = fabsf(hp[i * HW] - g);
        if (i == 0 || dist < best) { best = dist; nearest = i; }      // first minimum, like torch.min
    }
#pragma unroll
    for (int j = 0; j < D; ++j) a[j] = logf((j == nearest ? 1.0f : 0.0f) + 1e-12f);

    float uh[kMaxIters][D], vh[kMaxIters][D];      // potentials after every iteration (reverse sweep)
    float u[D], v[D];
#pragma unroll
    for (int i = 0; i < D; ++i) u[i] = 0.0f;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < D; ++i) m = fmaxf(m, fabsf((float)(i - j)) * inv_eps + u[i]);
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) s += expf(fabsf((float)(i - j)) * inv_eps + u[i] - m);
            v[j] = a[j] - (m + logf(s));
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < D; ++j) m = fmaxf(m, fabsf((float)(i - j)) * inv_eps + v[j]);
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) s += expf(fabsf((float)(i - j)) * inv_eps + v[j] - m);
            u[i] = bl[i] - (m + logf(s));
        }
#pragma unroll
        for (int i = 0; i < D; ++i) { uh[t][i] = u[i]; vh[t][i] = v[i]; }
    }
    if (iters == 0) {
#pragma unroll
        for (int j = 0; j < D; ++j) v[j] = 0.0f;
    }
    // loss and the gradients of the final plan
    float loss = 0.0f, du[D], dv[D], db[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { du[i] = 0.0f; dv[i] = 0.0f; db[i] = 0.0f; }
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
        float ut[D], vt[D], up[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            ut[i] = uh[t][i];
            vt[i] = vh[t][i];
            up[i] = t > 0 ? uh[t - 1][i] : 0.0f;
        }
        // u_i = b_i - LSE_j(K_ij + v_j):  softmax_ij = exp(K_ij + v_j + u_i - b_i)
#pragma unroll
        for (int i = 0; i < D; ++i) db[i] += du[i];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < D; ++i) acc += du[i] * expf(fabsf((float)(i - j)) * inv_eps + vt[j] + ut[i] - bl[i]);
            dv[j] -= acc;
        }
        // v_j = a_j - LSE_i(K_ij + u'_i) with u' the previous iterate:  softmax_ij = exp(K_ij + u'_i + v_j - a_j)
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < D; ++j) acc += dv[j] * expf(fabsf((float)(i - j)) * inv_eps + up[i] + vt[j] - a[j]);
            du[i] = -acc;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) dv[j] = 0.0f;
    }
    float* jp = jac + b * D * HW + q;
#pragma unroll
    for (int i = 0; i < D; ++i) jp[i * HW] = db[i] / (pred[i] + 1e-12f);
}

}  // namespace

// attn, hypo [B,D,HW]; gt [B,HW] -> loss_pix [B,HW], jac [B,D,HW] = d loss_pix / d attn.  D in {2,...,8}, iters <= 16.
extern "C" int mvster_sinkhorn(const float* attn, const float* hypo, const float* gt, float* loss_pix, float* jac, int B,
                               int D, long HW, int iters, float eps, void* stream) {
    if (!attn || !hypo || !gt || !loss_pix || !jac) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0 || iters < 0 || !(eps > 0.0f)) return MVSTER_ERR_SHAPE;
    if (iters > kMaxIters) return MVSTER_ERR_UNSUPPORTED;
    const long n = (long)B * HW;
    dim3 grid((unsigned)((n + 127) / 128)), block(128);
    hipStream_t s = (hipStream_t)stream;
    const float inv_eps = 1.0f / eps;
#define MV_S(D_) if (D == D_) { hipLaunchKernelGGL(sinkhorn_kernel<D_>, grid, block, 0, s, attn, hypo, gt, loss_pix, jac, B, HW, iters, inv_eps); return mv_check_launch(); }
    MV_S(2) MV_S(3) MV_S(4) MV_S(5) MV_S(6) MV_S(7) MV_S(8)
#undef MV_S
    return MVSTER_ERR_UNSUPPORTED;
}
