// Small fused kernels of the training step's "glue": what the reference writes as chains of tensor expressions around its
// convolutions (each a launch of a few microseconds inside the captured step; 258 such launches = 1.7 ms of a 15 ms step in
// round 5, profiles/r05_f_train_categories.txt):
//   * one stage of MVS4net_loss around the Sinkhorn term, forward (masked means, the weighted sum over the stages) and
//     backward (models/MVS4Net.py:126-153)
//   * the monocular head's disparity -> depth map, forward and backward (models/mvs4net_utils.py:858-866)
//   * nearest x2 up-sampling + channel concatenation in front of the head's 3x3 convolution (:854-857)
//   * the composed weights of the re-associated finest FPN level (train_ops.fpn_fine_level), forward and backward
//   * the Adam update of every parameter in two launches (train_mvs4.py:367, torch.optim.Adam)
// All latency-bound: one pass over [B,H,W]-sized planes or a few KB of parameters.
#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// Stage loss.  planes [2][n]: valid (mask > 0.5), valid * sign(mono - gt); partial [nblk][4]: per-workgroup sums of
// valid, valid*|mono-gt|, valid*out_of_range, valid*loss_pix (fixed order inside a workgroup: wave shuffles, then the four
// waves through LDS); out [6] = n, l1, ratio, ot, weighted, total (see mvster_stage_loss_fwd).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256) stage_loss_terms2_kernel(const float* __restrict__ hypo, const float* __restrict__ gt,
                                                                const float* __restrict__ mask,
                                                                const float* __restrict__ loss_pix,
                                                                const float* __restrict__ mono, float* __restrict__ planes,
                                                                float* __restrict__ partial, int B, int D, long HW,
                                                                int inverse) {
    __shared__ float red[4][4];
    const long n = (long)B * HW;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
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
        planes[p] = valid ? 1.0f : 0.0f;
        planes[n + p] = valid ? sg : 0.0f;
        // (a non-finite OT loss of a masked-out pixel -- ground-truth depth 0 -- stays out of the sum)
        s[0] += valid ? 1.0f : 0.0f;
        s[1] += valid ? l1 : 0.0f;
        s[2] += (valid && !inside) ? 1.0f : 0.0f;
        s[3] += valid ? loss_pix[p] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = wave_sum(s[j]);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][j] = s[j];
    }
    __syncthreads();
    if (threadIdx.x < 4) partial[(long)blockIdx.x * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void __launch_bounds__(256) stage_loss_finish_kernel(const float* __restrict__ partial, int nblk,
                                                                const float* __restrict__ total_in, float* __restrict__ out,
                                                                float w_l1, float w_ot, float w_stage) {
    __shared__ double red[256];
    const int col = threadIdx.x & 3, lane = threadIdx.x >> 2;          // 64 lanes per column
    double s = 0.0;
    for (int i = lane; i < nblk; i += 64) s += (double)partial[(long)i * 4 + col];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
        if (lane < w) red[threadIdx.x] += red[threadIdx.x + w * 4];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float n = (float)red[0];
        const float l1 = (float)red[1] / n, ratio = (float)red[2] / n, ot = (float)red[3] / n;
        // stage_lw * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot), rounded after every operation like the tensor expression
        const float weighted = __fmul_rn(w_stage, __fadd_rn(__fmul_rn(w_l1, l1), __fmul_rn(w_ot, ot)));
        out[0] = n; out[1] = l1; out[2] = ratio; out[3] = ot; out[4] = weighted;
        out[5] = __fadd_rn(total_in ? total_in[0] : 0.0f, weighted);
    }
}

// g_attn [B,D,HW] = valid * (c_ot / n) * jac (0 where the factor is 0: a non-finite Jacobian of a masked-out pixel stays
// out), g_mono [B,HW] = valid*sign * (c_l1 / n); c_x = g_x + g_total * w_x (either pointer may be null).
__global__ void __launch_bounds__(256) stage_loss_bwd_kernel(const float* __restrict__ jac, const float* __restrict__ planes,
                                                             const float* __restrict__ out, const float* __restrict__ g_total,
                                                             const float* __restrict__ g_l1, const float* __restrict__ g_ot,
                                                             float w_l1, float w_ot, float* __restrict__ g_attn,
                                                             float* __restrict__ g_mono, int B, int D, long HW) {
    const long n = (long)B * HW;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float cnt = out[0];
    const float gt_ = g_total ? g_total[0] : 0.0f;
    const float c_ot = (g_ot ? g_ot[0] : 0.0f) + gt_ * w_ot, c_l1 = (g_l1 ? g_l1[0] : 0.0f) + gt_ * w_l1;
    const float valid = planes[p];
    if (g_attn) {
        const long b = p / HW, q = p - b * HW;
        const float w = valid * (c_ot / cnt);
        for (int d = 0; d < D; ++d) {
            const long i = (b * D + d) * HW + q;
            g_attn[i] = w != 0.0f ? jac[i] * w : 0.0f;
        }
    }
    if (g_mono) g_mono[p] = planes[n + p] * (c_l1 / cnt);
}

// ---------------------------------------------------------------------------------------------------------------------
// Monocular head: depth = 1 / (lo + (hi - lo) * sigmoid(z)), lo = 1/d_max[b], hi = 1/d_min[b]  (mvs4net_utils.py:858-866)
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mono_depth_fwd_kernel(const float* __restrict__ z, const float* __restrict__ dmin,
                                                             const float* __restrict__ dmax, float* __restrict__ depth,
                                                             float* __restrict__ sig, int B, long HW) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)B * HW) return;
    const int b = (int)(p / HW);
    const float lo = 1.0f / dmax[b], hi = 1.0f / dmin[b];
    const float s = 1.0f / (1.0f + expf(-z[p]));
    sig[p] = s;
    depth[p] = 1.0f / (lo + (hi - lo) * s);
}

// d depth / d z = -depth^2 * (hi - lo) * s (1 - s)
__global__ void __launch_bounds__(256) mono_depth_bwd_kernel(const float* __restrict__ g, const float* __restrict__ depth,
                                                             const float* __restrict__ sig, const float* __restrict__ dmin,
                                                             const float* __restrict__ dmax, float* __restrict__ gz, int B,
                                                             long HW) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)B * HW) return;
    const int b = (int)(p / HW);
    const float lo = 1.0f / dmax[b], hi = 1.0f / dmin[b];
    const float d = depth[p], s = sig[p];
    gz[p] = (g[p] * -(d * d)) * (hi - lo) * ((1.0f - s) * s);
}

// ---------------------------------------------------------------------------------------------------------------------
// out [NB,H,W,Ca+Cb] = concat(nearest x2 of a [NB,H/2,W/2,Ca], b [NB,H,W,Cb]) and its adjoint; 4 channels per thread.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) upcat_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ out, int NB, int H, int W, int Ca, int Cb) {
    const int q = (Ca + Cb) >> 2;
    const long total = (long)NB * H * W * q;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % q) * 4;
    const long pix = i / q;
    const int x = (int)(pix % W);
    const long t = pix / W;
    const int y = (int)(t % H);
    const long nb = t / H;
    f32x4 v;
    if (c4 < Ca) v = ld4(a + ((nb * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * Ca + c4);
    else v = ld4(b + pix * Cb + (c4 - Ca));
    st4(out + pix * (Ca + Cb) + c4, v);
}

// ga [NB,H/2,W/2,Ca] = sum of the four children's first Ca channels; gb [NB,H,W,Cb] = the other channels
__global__ void __launch_bounds__(256) upcat_bwd_kernel(const float* __restrict__ g, float* __restrict__ ga,
                                                        float* __restrict__ gb, int NB, int H, int W, int Ca, int Cb) {
    const int C = Ca + Cb;
    const long na = (long)NB * (H >> 1) * (W >> 1) * (Ca >> 2), nb_ = (long)NB * H * W * (Cb >> 2);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < na) {
        const int qa = Ca >> 2;
        const int c4 = (int)(i % qa) * 4;
        const long pix = i / qa;
        const int x = (int)(pix % (W >> 1));
        const long t = pix / (W >> 1);
        const int y = (int)(t % (H >> 1));
        const long n = t / (H >> 1);
        const float* p = g + ((n * H + 2 * y) * W + 2 * x) * C + c4;
        f32x4 v = ld4(p);
        v += ld4(p + C);
        v += ld4(p + (long)W * C);
        v += ld4(p + (long)W * C + C);
        st4(ga + pix * Ca + c4, v);
    } else if (i - na < nb_) {
        const long j = i - na;
        const int qb = Cb >> 2;
        const int c4 = (int)(j % qb) * 4;
        const long pix = j / qb;
        st4(gb + pix * Cb + c4, ld4(g + pix * C + Ca + c4));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Composed weights of the re-associated finest FPN level (train_ops.fpn_fine_level): with wo [CO,CM,3,3] (out4.weight),
// wi [CM,CI] (inner3.weight), bi [CM] (inner3.bias):
//   wg [9*CO, CM]    row tap*CO + o = wo[o, :, tap]                          (the 1x1 product in front of the gather-sum)
//   wc [CO, CI, 3,3] = sum_c wo[o,c,tap] wi[c,i]                             (the 3x3 conv of the fine trunk map)
//   vb [9, CO]       = sum_c wo[o,c,tap] bi[c]                               (the gather-sum's per-tap bias terms)
// and the adjoint: g_wo = g_wg (permuted) + sum_i g_wc wi + g_vb bi, g_wi [CM,CI] = sum_{o,tap} wo g_wc,
// g_bi [CM] = sum_{o,tap} wo g_vb.  A few workgroups; a few thousand multiply-adds each.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fine_weights_fwd_kernel(const float* __restrict__ wo, const float* __restrict__ wi,
                                                               const float* __restrict__ bi, float* __restrict__ wg,
                                                               float* __restrict__ wc, float* __restrict__ vb, int CO, int CM,
                                                               int CI) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 9 * CO * CM; i += gridDim.x * 256) {             // wg[(tap*CO + o)*CM + c]
        const int c = i % CM, r = i / CM, o = r % CO, tap = r / CO;
        wg[i] = wo[(o * CM + c) * 9 + tap];
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < CO * CI * 9; i += gridDim.x * 256) {             // wc[(o*CI + ci)*9 + tap]
        const int tap = i % 9, ci = (i / 9) % CI, o = i / (9 * CI);
        float s = 0.0f;
        for (int c = 0; c < CM; ++c) s = fmaf(wo[(o * CM + c) * 9 + tap], wi[c * CI + ci], s);
        wc[i] = s;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 9 * CO; i += gridDim.x * 256) {                  // vb[tap*CO + o]
        const int o = i % CO, tap = i / CO;
        float s = 0.0f;
        for (int c = 0; c < CM; ++c) s = fmaf(wo[(o * CM + c) * 9 + tap], bi[c], s);
        vb[i] = s;
    }
}

__global__ void __launch_bounds__(256) fine_weights_bwd_kernel(const float* __restrict__ wo, const float* __restrict__ wi,
                                                               const float* __restrict__ bi, const float* __restrict__ g_wg,
                                                               const float* __restrict__ g_wc, const float* __restrict__ g_vb,
                                                               float* __restrict__ g_wo, float* __restrict__ g_wi,
                                                               float* __restrict__ g_bi, int CO, int CM, int CI) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < CO * CM * 9; i += gridDim.x * 256) {             // g_wo[(o*CM + c)*9 + tap]
        const int tap = i % 9, c = (i / 9) % CM, o = i / (9 * CM);
        float s = g_wg ? g_wg[((tap * CO + o)) * CM + c] : 0.0f;
        if (g_wc)
            for (int ci = 0; ci < CI; ++ci) s = fmaf(g_wc[(o * CI + ci) * 9 + tap], wi[c * CI + ci], s);
        if (g_vb) s = fmaf(g_vb[tap * CO + o], bi[c], s);
        g_wo[i] = s;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < CM * CI; i += gridDim.x * 256) {                 // g_wi[c*CI + ci]
        const int ci = i % CI, c = i / CI;
        float s = 0.0f;
        if (g_wc)
            for (int o = 0; o < CO; ++o)
                for (int tap = 0; tap < 9; ++tap) s = fmaf(wo[(o * CM + c) * 9 + tap], g_wc[(o * CI + ci) * 9 + tap], s);
        g_wi[i] = s;
    }
    for (int c = blockIdx.x * 256 + threadIdx.x; c < CM; c += gridDim.x * 256) {
        float s = 0.0f;
        if (g_vb)
            for (int o = 0; o < CO; ++o)
                for (int tap = 0; tap < 9; ++tap) s = fmaf(wo[(o * CM + c) * 9 + tap], g_vb[tap * CO + o], s);
        g_bi[c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam semantics, L2 weight decay, no amsgrad) over a list of tensors handed over as KERNEL ARGUMENTS
// (the gradients are fresh allocations every step, so a device-side pointer table would have to be re-uploaded; arguments
// are copied at launch and recorded with the launch in a captured step).  The moments live in one flat buffer each.
// step counter: read from step_in (the number of updates done so far), written (+1) to step_out by block 0 -- two cells,
// so that a second launch of the same step reads what the first wrote (see mvster_fused_adam).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kAdamMaxTensors = 128;     // (24 + 4 bytes each: the argument block stays under 4 KB)
struct AdamTensor {
    float* param;
    const float* grad;
    int state_off;        // element offset of this tensor's moments in the flat buffers
    int n;
};
struct AdamArgs {
    AdamTensor t[kAdamMaxTensors];
    int first_block[kAdamMaxTensors];     // prefix sum of ceil(n / 1024)
    int count;
};

__global__ void __launch_bounds__(256) fused_adam_kernel(const AdamArgs a, float* __restrict__ exp_avg,
                                                         float* __restrict__ exp_avg_sq, const float* __restrict__ step_in,
                                                         float* __restrict__ step_out, int bump,
                                                         const float* __restrict__ lr_cell, double beta1, double beta2,
                                                         double eps, double weight_decay) {
    int lo = 0, hi = a.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const AdamTensor t = a.t[lo];
    const float step = step_in[0] + (bump ? 1.0f : 0.0f);
    if (blockIdx.x == 0 && threadIdx.x == 0) step_out[0] = step;
    // the scalars in double, like torch's fused kernel (its betas / eps / step size are doubles that promote the element math)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = (double)lr_cell[0] / bc1;
    const double bc2_sqrt = sqrt(bc2);
    const int base = ((int)blockIdx.x - a.first_block[lo]) * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = base + j * 256 + threadIdx.x;
        if (i >= t.n) break;
        const float p = t.param[i];
        float g = t.grad[i];
        float m = exp_avg[t.state_off + i], v = exp_avg_sq[t.state_off + i];
        if (weight_decay != 0.0) g = (float)((double)g + weight_decay * (double)p);
        m = (float)((double)m + (1.0 - beta1) * ((double)g - (double)m));                 // lerp(m, g, 1 - beta1)
        v = (float)(beta2 * (double)v + (1.0 - beta2) * (double)g * (double)g);
        const double denom = (double)sqrtf(v) / bc2_sqrt + eps;
        t.param[i] = (float)((double)p - step_size * (double)m / denom);
        exp_avg[t.state_off + i] = m;
        exp_avg_sq[t.state_off + i] = v;
    }
}

}  // namespace

// One stage of MVS4net_loss around the OT term, forward: hypo [B,D,HW] (D >= 3), gt, mask (float, > 0.5 = valid),
// loss_pix [B,HW] (from mvster_sinkhorn*), mono [B,HW] or null ->
//   planes  [2][B*HW]   valid, valid * sign(mono - gt)            (for mvster_stage_loss_bwd)
//   partial [mvster_stage_loss_slots(B*HW)][4]                      (scratch)
//   out     [6] = number of valid pixels, l1 = mean |mono - gt| (0 without mono), out-of-range ratio, ot = mean OT loss,
//                 weighted = w_stage * (w_l1 * l1 + w_ot * ot), total = total_in[0] (0 if null) + weighted
// Two launches (per-workgroup partial sums, then their sum in a fixed order in fp64): deterministic.
extern "C" int mvster_stage_loss_slots(long n) {
    const long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int mvster_stage_loss_fwd(const float* hypo, const float* gt, const float* mask, const float* loss_pix,
                                     const float* mono, const float* total_in, float* planes, float* partial, float* out, int B,
                                     int D, long HW, int inverse_depth, float w_l1, float w_ot, float w_stage, void* stream) {
    if (!hypo || !gt || !mask || !loss_pix || !planes || !partial || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0 || D < 3) return MVSTER_ERR_SHAPE;
    const int nblk = mvster_stage_loss_slots((long)B * HW);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stage_loss_terms2_kernel, dim3(nblk), dim3(256), 0, s, hypo, gt, mask, loss_pix, mono, planes, partial, B, D,
                       HW, inverse_depth ? 1 : 0);
    hipLaunchKernelGGL(stage_loss_finish_kernel, dim3(1), dim3(256), 0, s, partial, nblk, total_in, out, w_l1, w_ot, w_stage);
    return mv_check_launch();
}

// Backward of the above: jac [B,D,HW] (d loss_pix / d attn from mvster_sinkhorn*), planes and out as written by the
// forward; g_total / g_l1 / g_ot: device scalars (null = 0), the gradients of out[5], out[1], out[3]; w_l1 = w_stage *
// l1ot_lw[0], w_ot = w_stage * l1ot_lw[1] -> g_attn [B,D,HW] (or null), g_mono [B,HW] (or null).  One launch.
extern "C" int mvster_stage_loss_bwd(const float* jac, const float* planes, const float* out, const float* g_total,
                                     const float* g_l1, const float* g_ot, float w_l1, float w_ot, float* g_attn, float* g_mono,
                                     int B, int D, long HW, void* stream) {
    if (!planes || !out || (g_attn && !jac)) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0 || D < 1) return MVSTER_ERR_SHAPE;
    const long n = (long)B * HW;
    hipLaunchKernelGGL(stage_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, jac, planes, out,
                       g_total, g_l1, g_ot, w_l1, w_ot, g_attn, g_mono, B, D, HW);
    return mv_check_launch();
}

// depth [B,HW] = 1 / (1/d_max[b] + (1/d_min[b] - 1/d_max[b]) * sigmoid(z)), sig [B,HW] = sigmoid(z) (kept for the backward)
extern "C" int mvster_mono_depth_fwd(const float* z, const float* dmin, const float* dmax, float* depth, float* sig, int B,
                                     long HW, void* stream) {
    if (!z || !dmin || !dmax || !depth || !sig) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0) return MVSTER_ERR_SHAPE;
    const long n = (long)B * HW;
    hipLaunchKernelGGL(mono_depth_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, dmin, dmax,
                       depth, sig, B, HW);
    return mv_check_launch();
}

extern "C" int mvster_mono_depth_bwd(const float* g, const float* depth, const float* sig, const float* dmin, const float* dmax,
                                     float* gz, int B, long HW, void* stream) {
    if (!g || !depth || !sig || !dmin || !dmax || !gz) return MVSTER_ERR_NULL;
    if (B <= 0 || HW <= 0) return MVSTER_ERR_SHAPE;
    const long n = (long)B * HW;
    hipLaunchKernelGGL(mono_depth_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, depth, sig,
                       dmin, dmax, gz, B, HW);
    return mv_check_launch();
}

// out [NB,H,W,Ca+Cb] = concat(nearest x2 up-sampling of a [NB,H/2,W/2,Ca], b [NB,H,W,Cb]); H, W even, Ca, Cb multiples of 4
extern "C" int mvster_upcat_fwd(const float* a, const float* b, float* out, int NB, int H, int W, int Ca, int Cb, void* stream) {
    if (!a || !b || !out) return MVSTER_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0 || ((H | W) & 1)) return MVSTER_ERR_SHAPE;
    if (Ca <= 0 || Cb <= 0 || ((Ca | Cb) & 3)) return MVSTER_ERR_UNSUPPORTED;
    const long total = (long)NB * H * W * ((Ca + Cb) >> 2);
    hipLaunchKernelGGL(upcat_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, NB, H,
                       W, Ca, Cb);
    return mv_check_launch();
}

// the adjoint: g [NB,H,W,Ca+Cb] -> ga [NB,H/2,W/2,Ca], gb [NB,H,W,Cb]
extern "C" int mvster_upcat_bwd(const float* g, float* ga, float* gb, int NB, int H, int W, int Ca, int Cb, void* stream) {
    if (!g || !ga || !gb) return MVSTER_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0 || ((H | W) & 1)) return MVSTER_ERR_SHAPE;
    if (Ca <= 0 || Cb <= 0 || ((Ca | Cb) & 3)) return MVSTER_ERR_UNSUPPORTED;
    const long total = (long)NB * (H >> 1) * (W >> 1) * (Ca >> 2) + (long)NB * H * W * (Cb >> 2);
    hipLaunchKernelGGL(upcat_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, ga, gb, NB, H,
                       W, Ca, Cb);
    return mv_check_launch();
}

// wo [CO,CM,3,3], wi [CM,CI], bi [CM] -> wg [9*CO,CM], wc [CO,CI,3,3], vb [9,CO] (see the kernel)
extern "C" int mvster_fine_weights_fwd(const float* wo, const float* wi, const float* bi, float* wg, float* wc, float* vb, int CO,
                                       int CM, int CI, void* stream) {
    if (!wo || !wi || !bi || !wg || !wc || !vb) return MVSTER_ERR_NULL;
    if (CO <= 0 || CM <= 0 || CI <= 0) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(fine_weights_fwd_kernel, dim3((9 * CO * CM + 255) / 256), dim3(256), 0, (hipStream_t)stream, wo, wi, bi, wg, wc, vb, CO, CM, CI);
    return mv_check_launch();
}

// gradients of wg / wc / vb (any may be null) -> g_wo [CO,CM,3,3], g_wi [CM,CI], g_bi [CM]
extern "C" int mvster_fine_weights_bwd(const float* wo, const float* wi, const float* bi, const float* g_wg, const float* g_wc,
                                       const float* g_vb, float* g_wo, float* g_wi, float* g_bi, int CO, int CM, int CI,
                                       void* stream) {
    if (!wo || !wi || !bi || !g_wo || !g_wi || !g_bi) return MVSTER_ERR_NULL;
    if (CO <= 0 || CM <= 0 || CI <= 0) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(fine_weights_bwd_kernel, dim3((9 * CO * CM + 255) / 256), dim3(256), 0, (hipStream_t)stream, wo, wi, bi, g_wg, g_wc, g_vb, g_wo,
                       g_wi, g_bi, CO, CM, CI);
    return mv_check_launch();
}

// Adam update of `count` tensors: params / grads = arrays of `count` device pointers (host arrays), sizes / state_offs =
// host int arrays (elements; offsets into exp_avg / exp_avg_sq).  step_cells [2] floats on the device: cell 0 holds the
// number of updates done so far and holds it + 1 afterwards (cell 1 is scratch).  lr: one float ON THE DEVICE (a learning-
// rate schedule then reaches a captured step: the host rewrites the cell between replays).  Launches of at most 128 tensors each;
// with an odd number of launches a one-thread copy brings the count back to cell 0 -- callers see cell 0 only.
extern "C" int mvster_fused_adam(const void* const* params, const void* const* grads, const int* sizes, const int* state_offs,
                                 int count, float* exp_avg, float* exp_avg_sq, float* step_cells, const float* lr, double beta1,
                                 double beta2, double eps, double weight_decay, void* stream) {
    if (!params || !grads || !sizes || !state_offs || !exp_avg || !exp_avg_sq || !step_cells || !lr) return MVSTER_ERR_NULL;
    if (count <= 0) return MVSTER_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int launches = (count + kAdamMaxTensors - 1) / kAdamMaxTensors;
    // every launch reads `in`, writes `out`; the first one adds 1.  With an even number of launches the count ends in cell 0;
    // with an odd number the roles start swapped... not possible (the input IS cell 0), so a last empty-handed launch
    // (one tensor of zero elements) moves it back.
    int in = 0;
    for (int l = 0; l < launches; ++l) {
        AdamArgs a;
        a.count = 0;
        int blocks = 0;
        for (int i = l * kAdamMaxTensors; i < count && a.count < kAdamMaxTensors; ++i) {
            if (!params[i] || !grads[i] || sizes[i] < 0) return MVSTER_ERR_NULL;
            a.t[a.count] = AdamTensor{(float*)params[i], (const float*)grads[i], state_offs[i], sizes[i]};
            a.first_block[a.count] = blocks;
            blocks += (sizes[i] + 1023) / 1024;
            ++a.count;
        }
        if (blocks == 0) blocks = 1;
        hipLaunchKernelGGL(fused_adam_kernel, dim3(blocks), dim3(256), 0, s, a, exp_avg, exp_avg_sq, step_cells + in,
                           step_cells + (in ^ 1), l == 0 ? 1 : 0, lr, beta1, beta2, eps, weight_decay);
        in ^= 1;
    }
    if (in == 1) {           // the count sits in cell 1: one more (empty) launch copies it to cell 0
        AdamArgs a;
        a.count = 1;
        a.t[0] = AdamTensor{nullptr, nullptr, 0, 0};
        a.first_block[0] = 0;
        hipLaunchKernelGGL(fused_adam_kernel, dim3(1), dim3(256), 0, s, a, exp_avg, exp_avg_sq, step_cells + 1, step_cells, 0, lr,
                           beta1, beta2, eps, weight_decay);
    }
    return mv_check_launch();
}
