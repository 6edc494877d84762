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
// slot of `partial` [nblk][2][C]; the host adds the slots (deterministic).
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
// mapping and per-workgroup partial slots as the backward reduction; the host finishes in fp64.
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

// Finish the statistics on the device (two launches instead of ~20 tiny tensor ops per BatchNorm call): reduce the
// per-workgroup partial sums in fp64 and form mean / biased variance / rstd / scale / shift per (group, channel)
// [bn_finalize_kernel, one workgroup of 1024 threads per group: thread t owns channel t % C and every (1024 / C)-th
// partial slot], then apply the running-average updates of the groups one after the other (what `groups` sequential
// module calls would do) [bn_running_kernel, one thread per channel].
__global__ void __launch_bounds__(1024) bn_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ x,
                                                           const float* __restrict__ weight, const float* __restrict__ bias,
                                                           float* __restrict__ out, long rows, int C, int groups, int nblk,
                                                           float eps) {
    __shared__ double red[2][1024];
    const int g = blockIdx.x;
    const int c = threadIdx.x % C, sl = threadIdx.x / C, nsl = 1024 / C;
    const float* pg = partial + (long)g * nblk * 2 * C;
    double s1 = 0.0, s2 = 0.0;
    for (int n = sl; n < nblk; n += nsl) {
        s1 += (double)pg[((long)n * 2 + 0) * C + c];
        s2 += (double)pg[((long)n * 2 + 1) * C + c];
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int j = 1; j < nsl; ++j) { s1 += red[0][j * C + c]; s2 += red[1][j * C + c]; }
        const double m1 = s1 / (double)rows, m2 = s2 / (double)rows;
        const float mean = x[(long)g * rows * C + c] + (float)m1;
        float var = (float)(m2 - m1 * m1);
        var = var > 0.0f ? var : 0.0f;
        const float rstd = 1.0f / sqrtf(var + eps);
        const float scale = weight[c] * rstd;
        float* o = out + (long)g * C + c;
        const long gs = (long)groups * C;
        o[0] = mean; o[gs] = var; o[2 * gs] = rstd; o[3 * gs] = scale; o[4 * gs] = bias[c] - mean * scale;
    }
}

__global__ void bn_running_kernel(const float* __restrict__ out, float* __restrict__ running_mean,
                                  float* __restrict__ running_var, long rows, int C, int groups, float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float unbias = (float)rows / (float)(rows > 1 ? rows - 1 : 1);
    float rm = running_mean[c], rv = running_var[c];
    for (int g = 0; g < groups; ++g) {
        rm = (1.0f - momentum) * rm + momentum * out[(long)g * C + c];
        rv = (1.0f - momentum) * rv + momentum * (out[((long)groups + g) * C + c] * unbias);
    }
    running_mean[c] = rm;
    running_var[c] = rv;
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

// partial [nblk][2][C]; nblk is returned by mvster_bn_blocks(rows, C)
extern "C" int mvster_bn_blocks(long rows, int C) {
    if (check(rows, C)) return 0;
    return blocks_for(rows * (C / 4));
}

// partial [groups][nblk][2][C]: sums of (x - x[first row of the group]) and of its square
extern "C" int mvster_bn_stats(const float* x, float* partial, long rows, int C, int groups, void* stream) {
    if (!x || !partial) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(blocks_for(n4), groups), dim3(256), 0, (hipStream_t)stream, x, partial, n4, C);
    return mv_check_launch();
}

// out [5][groups][C] = mean, biased var, rstd, scale = gamma*rstd, shift = beta - mean*scale; running_mean / running_var
// (optional, [C]) receive the `groups` exponential-average updates in group order (unbiased variance, torch semantics).
// partial as written by mvster_bn_stats with nblk = mvster_bn_blocks(rows, C).
extern "C" int mvster_bn_finalize(const float* partial, const float* x, const float* weight, const float* bias,
                                  float* running_mean, float* running_var, float* out, long rows, int C, int groups, float eps,
                                  float momentum, void* stream) {
    if (!partial || !x || !weight || !bias || !out) return MVSTER_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const int nblk = blocks_for(rows * (C / 4));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(groups), dim3(1024), 0, (hipStream_t)stream, partial, x, weight, bias, out, rows,
                       C, groups, nblk, eps);
    if (running_mean)
        hipLaunchKernelGGL(bn_running_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, running_mean, running_var, rows, C,
                           groups, momentum);
    return mv_check_launch();
}

// partial [groups][nblk][2][C]
extern "C" int mvster_bn_relu_bwd_reduce(const float* x, const float* gy, const float* scale, const float* shift,
                                         const float* mean, const float* rstd, float* partial, long rows, int C, int relu,
                                         int groups, void* stream) {
    if (!x || !gy || !scale || !shift || !mean || !rstd || !partial) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(blocks_for(n4), groups), dim3(256), 0, (hipStream_t)stream, x, gy, scale,
                       shift, mean, rstd, partial, n4, C, relu);
    return mv_check_launch();
}

// sums [groups][2][C] = (sum g, sum g*xh) over the group's rows; dx [groups*rows, C]
extern "C" int mvster_bn_relu_bwd_apply(const float* x, const float* gy, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, const float* sums, float* dx, long rows, int C,
                                        int relu, int groups, void* stream) {
    if (!x || !gy || !scale || !shift || !mean || !rstd || !sums || !dx) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(blocks_for(n4), groups), dim3(256), 0, (hipStream_t)stream, x, gy, scale,
                       shift, mean, rstd, sums, dx, n4, C, relu, 1.0f / (float)rows);
    return mv_check_launch();
}
