// Training-mode BatchNorm + ReLU on channels-last activations [rows, C] (C in {4,...,64}, multiple of 4):
// the elementwise half of the reference's conv -> BatchNorm -> ReLU blocks (models/mvs4net_utils.py:116-123,
// :224-251) under autograd.  HBM-bound streaming kernels; the batch statistics themselves come from the host
// side (torch.var_mean, one pass).  Forward 1 read + 1 write; backward 4 reads + 1 write (PyTorch's autograd
// over the unfused ops makes ~19 passes).
//   y  = relu(x * scale + shift)                  scale = gamma * rstd, shift = beta - mean * scale
//   g  = gy * (y > 0)            xh = (x - mean) * rstd
//   dbeta = sum g     dgamma = sum g * xh         dx = scale * (g - dbeta/N - xh * dgamma/N)
// Every thread owns one float4 column group (blockDim*4 is a multiple of C), so per-channel sums stay in
// registers along the grid-stride loop and are reduced once per workgroup through LDS into a per-workgroup
// slot of `partial` [groups][nblk][2][C].  A small second kernel adds the slots in a fixed order and in fp64
// (deterministic) and writes the finished statistics / sums.  (Finishing in the last workgroup of the first kernel
// instead -- a ticket counter -- needs an agent-scope fence per workgroup, which on gfx950 writes the XCD's L2 back:
// measured +5 ms per training step.)
#include "common.hpp"

namespace {

__device__ __forceinline__ float bn_act(float x, float sc, float sh) { return fmaf(x, sc, sh); }

// 1 where the gradient passes the ReLU.  Kept as a multiplicative mask: hipcc 7.2 if-converts
// `cond ? g : 0.0f` on a just-loaded g into "g = 0; if (cond) {}" in the apply kernel (wrong code, caught by
// tests/test_gpu_train.py::test_batch_norm_cl_matches_torch).
__device__ __forceinline__ float relu_mask(int relu, float act) { return (relu == 0 || act > 0.0f) ? 1.0f : 0.0f; }

// blockIdx.y = statistics group (the reference normalises every view's batch separately): group g owns rows
// [g*rows, (g+1)*rows) of x and row g of the per-channel parameter arrays.
__global__ void __launch_bounds__(256) bn_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ skip,
                                                          float* __restrict__ y, long n4, int C, int relu) {
    x += (long)blockIdx.y * n4 * 4; y += (long)blockIdx.y * n4 * 4;
    if (skip) skip += (long)blockIdx.y * n4 * 4;
    scale += blockIdx.y * C; shift += blockIdx.y * C;
    const int q = C >> 2;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 sc = ld4(scale + cg), sh = ld4(shift + cg);
    for (; i < n4; i += stride) {
        const f32x4 v = ld4(x + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = bn_act(v[j], sc[j], sh[j]);
            o[j] = relu ? fmaxf(t, 0.0f) : t;
        }
        if (skip) o += ld4(skip + i * 4);        // U-Net skip connection, added after the activation (mvs4net_utils.py:893-895)
        st4(y + i * 4, o);
    }
}

// Batch statistics, one pass: per (group, channel) sums of (x - p) and (x - p)^2 with the pivot p = the group's
// first row (keeps the E[d^2] - E[d]^2 subtraction well conditioned whatever the channel's mean is).  Same thread
// mapping and per-workgroup partial slots as the backward reduction.
__global__ void __launch_bounds__(256) bn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial, long n4, int C) {
    __shared__ float red[256][8];
    x += (long)blockIdx.y * n4 * 4;
    partial += (long)blockIdx.y * gridDim.x * 2 * C;
    const int q = C >> 2;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 pv = ld4(x + cg);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (; i < n4; i += stride) {
        const f32x4 v = ld4(x + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[j] - pv[j];
            s1[j] += d;
            s2[j] = fmaf(d, d, s2[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][4 + j] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < q * 8) {
        const int grp = threadIdx.x >> 3, val = threadIdx.x & 7;
        float s = 0.0f;
        for (int t = grp; t < 256; t += q) s += red[t][val];
        partial[((long)blockIdx.x * 2 + (val >> 2)) * C + grp * 4 + (val & 3)] = s;
    }
}

// The finishing kernels: workgroup b owns channels 4b..4b+3 (8 columns: both slot rows), 128 threads per column
// strided over the slots, fp64, LDS tree; the groups are walked one after the other so that whatever depends on their
// order -- the running averages, the sums over the groups -- is formed in registers by the column's first thread.
// (7 / 6 us per launch, 108 launches per training step.  Two rewrites -- 64 lanes per column meeting by shuffles and one
// LDS pass for all groups; then every slot load of 8 groups issued up front, 32 per thread -- measured 8.5 / 7.6 and
// 9.0 / 7.9 us: the slot loads are not what these launches wait for.)
constexpr int kFinishLanes = 128;

__device__ __forceinline__ double column_sum(const float* __restrict__ pg, int nblk, int C, double* red) {
    const int col = threadIdx.x & 7, lane = threadIdx.x >> 3;
    const int off = (col >> 2) * C + blockIdx.x * 4 + (col & 3);
    double s = 0.0;
    for (int n = lane; n < nblk; n += kFinishLanes) s += (double)pg[(long)n * 2 * C + off];
    __syncthreads();                       // (red is reused from group to group)
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = kFinishLanes / 2; w > 0; w >>= 1) {
        if (lane < w) red[threadIdx.x] += red[threadIdx.x + w * 8];
        __syncthreads();
    }
    return red[col];                       // every thread gets its column's total
}

// Statistics: out [5][groups][C] = mean, biased variance, rstd, scale, shift; then the running-average updates of the
// groups one after the other (what `groups` sequential module calls would do: momentum, unbiased variance) and
// num_batches_tracked += groups.
struct BnFinalizeArgs {
    const float* partial; const float* x; const float* weight; const float* bias;
    float* running_mean; float* running_var; long* num_batches_tracked; float* out;
    long rows; int C, groups, nblk; float eps, momentum;
};

__global__ void __launch_bounds__(kFinishLanes * 8) bn_finalize_kernel(BnFinalizeArgs a) {
    __shared__ double red[kFinishLanes * 8];
    __shared__ double tot[8];
    const int C = a.C, c = blockIdx.x * 4 + (threadIdx.x & 3);
    const bool owner = threadIdx.x < 4, running = a.running_mean != nullptr;
    const float unbias = (float)a.rows / (float)(a.rows > 1 ? a.rows - 1 : 1);
    float rm = 0.0f, rv = 0.0f;
    if (owner && running) { rm = a.running_mean[c]; rv = a.running_var[c]; }
    for (int g = 0; g < a.groups; ++g) {
        const double t = column_sum(a.partial + (long)g * a.nblk * 2 * C, a.nblk, C, red);
        if (threadIdx.x < 8) tot[threadIdx.x] = t;
        __syncthreads();
        if (owner) {
            const double m1 = tot[threadIdx.x] / (double)a.rows, m2 = tot[4 + threadIdx.x] / (double)a.rows;
            const float mean = a.x[(long)g * a.rows * C + c] + (float)m1;
            float var = (float)(m2 - m1 * m1);
            var = var > 0.0f ? var : 0.0f;
            const float rstd = 1.0f / sqrtf(var + a.eps);
            const float scale = a.weight[c] * rstd;
            float* o = a.out + (long)g * C + c;
            const long gs = (long)a.groups * C;
            o[0] = mean; o[gs] = var; o[2 * gs] = rstd; o[3 * gs] = scale; o[4 * gs] = a.bias[c] - mean * scale;
            rm = (1.0f - a.momentum) * rm + a.momentum * mean;
            rv = (1.0f - a.momentum) * rv + a.momentum * (var * unbias);
        }
    }
    if (owner && running) { a.running_mean[c] = rm; a.running_var[c] = rv; }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.num_batches_tracked) *a.num_batches_tracked += a.groups;
}

// Backward sums: sums [groups][2][C] for the apply kernel and the parameter gradients summed over the groups,
// dbeta [C] = sum_g sum g_, dgamma [C] = sum_g sum g_*xh.
__global__ void __launch_bounds__(kFinishLanes * 8) bn_bwd_finish_kernel(const float* __restrict__ partial,
                                                                         float* __restrict__ sums, float* __restrict__ dgamma,
                                                                         float* __restrict__ dbeta, int C, int groups, int nblk) {
    __shared__ double red[kFinishLanes * 8];
    const int col = threadIdx.x & 7, c = blockIdx.x * 4 + (col & 3);
    double total = 0.0;
    for (int g = 0; g < groups; ++g) {
        const double t = column_sum(partial + (long)g * nblk * 2 * C, nblk, C, red);
        if (threadIdx.x < 8) sums[((long)g * 2 + (col >> 2)) * C + c] = (float)t;
        total += t;
    }
    if (threadIdx.x < 4) dbeta[c] = (float)total;
    else if (threadIdx.x < 8) dgamma[c] = (float)total;
}

__global__ void __launch_bounds__(256) bn_relu_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ partial,
                                                                 long n4, int C, int relu) {
    __shared__ float red[256][8];
    x += (long)blockIdx.y * n4 * 4; gy += (long)blockIdx.y * n4 * 4;
    scale += blockIdx.y * C; shift += blockIdx.y * C; mean += blockIdx.y * C; rstd += blockIdx.y * C;
    partial += (long)blockIdx.y * gridDim.x * 2 * C;
    const int q = C >> 2;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 sc = ld4(scale + cg), sh = ld4(shift + cg), mu = ld4(mean + cg), rs = ld4(rstd + cg);
    float sg[4] = {0.f, 0.f, 0.f, 0.f}, sgx[4] = {0.f, 0.f, 0.f, 0.f};
    for (; i < n4; i += stride) {
        const f32x4 v = ld4(x + i * 4), g4 = ld4(gy + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = g4[j] * relu_mask(relu, bn_act(v[j], sc[j], sh[j]));
            sg[j] += g;
            sgx[j] = fmaf(g, (v[j] - mu[j]) * rs[j], sgx[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { red[threadIdx.x][j] = sg[j]; red[threadIdx.x][4 + j] = sgx[j]; }
    __syncthreads();
    // thread t < q*8 sums one (column group, value) over the 256/q threads that share the column group
    if (threadIdx.x < q * 8) {
        const int grp = threadIdx.x >> 3, val = threadIdx.x & 7;
        float s = 0.0f;
        for (int t = grp; t < 256; t += q) s += red[t][val];
        const int c = grp * 4 + (val & 3);
        partial[((long)blockIdx.x * 2 + (val >> 2)) * C + c] = s;
    }
}

__global__ void __launch_bounds__(256) bn_relu_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ sums, float* __restrict__ dx,
                                                                long n4, int C, int relu, float inv_n) {
    x += (long)blockIdx.y * n4 * 4; gy += (long)blockIdx.y * n4 * 4; dx += (long)blockIdx.y * n4 * 4;
    scale += blockIdx.y * C; shift += blockIdx.y * C; mean += blockIdx.y * C; rstd += blockIdx.y * C;
    sums += blockIdx.y * 2 * C;
    const int q = C >> 2;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 sc = ld4(scale + cg), sh = ld4(shift + cg), mu = ld4(mean + cg), rs = ld4(rstd + cg);
    const f32x4 s0 = ld4(sums + cg), s1 = ld4(sums + C + cg);
    for (; i < n4; i += stride) {
        const f32x4 v = ld4(x + i * 4), g4 = ld4(gy + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = g4[j] * relu_mask(relu, bn_act(v[j], sc[j], sh[j]));
            const float xh = (v[j] - mu[j]) * rs[j];
            o[j] = sc[j] * (g - s0[j] * inv_n - xh * s1[j] * inv_n);
        }
        st4(dx + i * 4, o);
    }
}

int check(long rows, int C) {
    if (rows <= 0) return MVSTER_ERR_SHAPE;
    if (C < 4 || C > 64 || (C & (C - 1)) != 0) return MVSTER_ERR_UNSUPPORTED;
    return MVSTER_OK;
}

// Slots per group of `partial` for the two reductions: enough workgroups to stream at full rate, few enough that the
// finishing kernel adds them up in a few microseconds.
int slots_for(long rows, int C, int groups) {
    const long n4 = rows * (C / 4);
    long n = (n4 + 255) / 256;
    const long cap = groups > 1 ? 512 : 1024;
    if (n > cap) n = cap;
    return (int)(n < 1 ? 1 : n);
}

int blocks_for(long n4) {
    const long want = (n4 + 255) / 256;
    return (int)(want < 2048 ? want : 2048);
}

}  // namespace

// rows = rows PER GROUP; x, y [groups*rows, C]; scale, shift (mean, rstd) [groups, C]
extern "C" int mvster_bn_relu_fwd(const float* x, const float* scale, const float* shift, const float* skip, float* y,
                                  long rows, int C, int relu, int groups, void* stream) {
    if (!x || !scale || !shift || !y) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(blocks_for(n4), groups), dim3(256), 0, (hipStream_t)stream, x, scale, shift, skip,
                       y, n4, C, relu);
    return mv_check_launch();
}

// partial [groups][mvster_bn_slots(rows, C, groups)][2][C] floats (scratch of stats and bwd_reduce)
extern "C" int mvster_bn_slots(long rows, int C, int groups) {
    if (check(rows, C) || groups < 1 || groups > 65535) return 0;
    return slots_for(rows, C, groups);
}

// out [5][groups][C] = mean, biased var, rstd, scale = gamma*rstd, shift = beta - mean*scale; running_mean / running_var
// (optional, [C]) receive the `groups` exponential-average updates in group order (unbiased variance, torch semantics)
// and num_batches_tracked (optional, int64 on the device) += groups.  Two launches.
extern "C" int mvster_bn_stats(const float* x, const float* weight, const float* bias, float* running_mean,
                               float* running_var, long* num_batches_tracked, float* partial, float* out, long rows, int C,
                               int groups, float eps, float momentum, void* stream) {
    if (!x || !weight || !bias || !partial || !out) return MVSTER_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const int nblk = slots_for(rows, C, groups);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk, groups), dim3(256), 0, s, x, partial, rows * (C / 4), C);
    BnFinalizeArgs a{partial, x, weight, bias, running_mean, running_var, num_batches_tracked, out,
                     rows, C, groups, nblk, eps, momentum};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C / 4), dim3(kFinishLanes * 8), 0, s, a);
    return mv_check_launch();
}

// sums [groups][2][C] (for bwd_apply), dgamma [C], dbeta [C]; partial as for mvster_bn_stats.  Two launches.
extern "C" int mvster_bn_relu_bwd_reduce(const float* x, const float* gy, const float* scale, const float* shift,
                                         const float* mean, const float* rstd, float* partial, float* sums, float* dgamma,
                                         float* dbeta, long rows, int C, int relu, int groups, void* stream) {
    if (!x || !gy || !scale || !shift || !mean || !rstd || !partial || !sums || !dgamma || !dbeta) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    const int nblk = slots_for(rows, C, groups);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(nblk, groups), dim3(256), 0, s, x, gy, scale, shift, mean, rstd, partial,
                       n4, C, relu);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(C / 4), dim3(kFinishLanes * 8), 0, s, partial, sums, dgamma, dbeta, C, groups,
                       nblk);
    return mv_check_launch();
}

// sums [groups][2][C] = (sum g, sum g*xh) over the group's rows; dx [groups*rows, C].  frozen = 1: mean / rstd are
// constants (running statistics of a BatchNorm in eval mode), so dx = g * scale and the sums are not applied.
extern "C" int mvster_bn_relu_bwd_apply(const float* x, const float* gy, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, const float* sums, float* dx, long rows, int C,
                                        int relu, int groups, int frozen, void* stream) {
    if (!x || !gy || !scale || !shift || !mean || !rstd || !sums || !dx) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(blocks_for(n4), groups), dim3(256), 0, (hipStream_t)stream, x, gy, scale,
                       shift, mean, rstd, sums, dx, n4, C, relu, frozen ? 0.0f : 1.0f / (float)rows);
    return mv_check_launch();
}
