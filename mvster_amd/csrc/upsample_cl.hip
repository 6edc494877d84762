// Bilinear x2 (align_corners=True) up-sampling of channels-last feature maps and its adjoint: the FPN's top-down
// path in training (models/mvs4net_utils.py:488-496 under autograd).  PyTorch's NHWC backward scatters with atomics
// (1.4 ms for the 64-channel full-resolution map here); this one is a gather over the <= 6 x 6 output pixels that can
// touch an input pixel, weights from the same make_lerp() as the forward: no atomics, HBM-bound.
#include "common.hpp"

namespace {

// out [B, 2h, 2w, C] <- in [B, h, w, C]; one thread per (output pixel, 4 channels)
__global__ void __launch_bounds__(256) upsample2x_cl_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                                int h, int w, int C, FastDiv qd, FastDiv wd, FastDiv hd) {
    const int q = C >> 2, H = 2 * h, W = 2 * w;
    const unsigned total = (unsigned)B * H * W * q;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    unsigned c4u, xu, yu;
    const int b = (int)fdivmod(fdivmod(fdivmod(i, qd, c4u), wd, xu), hd, yu);      // divisors q, W = 2w, H = 2h
    const int c4 = (int)c4u, x = (int)xu, y = (int)yu;
    const mv::Lerp ly = mv::make_lerp(y, h, H), lx = mv::make_lerp(x, w, W);
    const float* base = in + (long)b * h * w * C + c4 * 4;
    const f32x4 v00 = ld4(base + ((long)ly.i0 * w + lx.i0) * C), v01 = ld4(base + ((long)ly.i0 * w + lx.i1) * C);
    const f32x4 v10 = ld4(base + ((long)ly.i1 * w + lx.i0) * C), v11 = ld4(base + ((long)ly.i1 * w + lx.i1) * C);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = mv::bilerp(ly, lx, v00[j], v01[j], v10[j], v11[j]);
    st4(out + (long)i * 4, o);
}

// weight with which output index o contributes to input index i along one axis
__device__ __forceinline__ float axis_weight(int o, int i, int in_size, int out_size) {
    const mv::Lerp l = mv::make_lerp(o, in_size, out_size);
    return (l.i0 == i ? l.w0 : 0.0f) + (l.i1 == i ? l.w1 : 0.0f);
}

// gin [B, h, w, C] <- gout [B, 2h, 2w, C]; one thread per (input pixel, 4 channels)
__global__ void __launch_bounds__(256) upsample2x_cl_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, int B,
                                                                int h, int w, int C, FastDiv qd, FastDiv wd, FastDiv hd) {
    const int q = C >> 2, H = 2 * h, W = 2 * w;
    const unsigned total = (unsigned)B * h * w * q;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    unsigned c4u, xu, yu;
    const int b = (int)fdivmod(fdivmod(fdivmod(i, qd, c4u), wd, xu), hd, yu);      // divisors q, w, h
    const int c4 = (int)c4u, xi = (int)xu, yi = (int)yu;
    // output rows / columns whose source coordinate lies within one pixel of (yi, xi): o * (in-1)/(out-1) in (i-1, i+1)
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int yo = 2 * yi - 2 + k, xo = 2 * xi - 2 + k;
        wy[k] = (yo >= 0 && yo < H) ? axis_weight(yo, yi, h, H) : 0.0f;
        wx[k] = (xo >= 0 && xo < W) ? axis_weight(xo, xi, w, W) : 0.0f;
    }
    const float* base = gout + (long)b * H * W * C + c4 * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 6; ++ky) {
        if (wy[ky] == 0.0f) continue;
        const int yo = 2 * yi - 2 + ky;
#pragma unroll
        for (int kx = 0; kx < 6; ++kx) {
            if (wx[kx] == 0.0f) continue;
            const int xo = 2 * xi - 2 + kx;
            const f32x4 g = ld4(base + ((long)yo * W + xo) * C);
            const float wgt = wy[ky] * wx[kx];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wgt, g[j], acc[j]);
        }
    }
    st4(gin + (long)i * 4, acc);
}

// nearest x2 (F.interpolate(scale_factor=2, mode="nearest"), the mono head's up-sampling, mvs4net_utils.py:858) and its
// adjoint (sum of the 2 x 2 block): out[y][x] = in[y/2][x/2]
__global__ void __launch_bounds__(256) upsample2x_nearest_cl_kernel(const float* __restrict__ in, float* __restrict__ out, int B,
                                                                    int h, int w, int C, int backward) {
    const int q = C >> 2, H = 2 * h, W = 2 * w;
    const int oh = backward ? h : H, ow = backward ? w : W;
    const long total = (long)B * oh * ow * q;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % q);
    long p = i / q;
    const int x = (int)(p % ow); p /= ow;
    const int y = (int)(p % oh);
    const int b = (int)(p / oh);
    if (!backward) {
        st4(out + i * 4, ld4(in + (((long)b * h + (y >> 1)) * w + (x >> 1)) * C + c4 * 4));
    } else {
        const float* g = in + (((long)b * H + 2 * y) * W + 2 * x) * C + c4 * 4;
        const f32x4 a = ld4(g), bq = ld4(g + C), c = ld4(g + (long)W * C), d = ld4(g + (long)W * C + C);
        st4(out + i * 4, (a + bq) + (c + d));
    }
}

}  // namespace

// in [B,h,w,C] -> out [B,2h,2w,C] (backward = 0), or gout [B,2h,2w,C] -> gin [B,h,w,C] (backward = 1); C % 4 == 0
extern "C" int mvster_upsample2x_nearest_cl(const float* in, float* out, int B, int h, int w, int C, int backward,
                                            void* stream) {
    if (!in || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3)) return MVSTER_ERR_SHAPE;
    const long total = (long)B * (backward ? 1 : 4) * h * w * (C / 4);
    if ((total + 255) / 256 >= (1L << 31)) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(upsample2x_nearest_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       out, B, h, w, C, backward);
    return mv_check_launch();
}

// in [B,h,w,C] -> out [B,2h,2w,C] (C % 4 == 0), F.interpolate(scale_factor=2, mode="bilinear", align_corners=True)
extern "C" int mvster_upsample2x_cl_fwd(const float* in, float* out, int B, int h, int w, int C, void* stream) {
    if (!in || !out) return MVSTER_ERR_NULL;
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3)) return MVSTER_ERR_SHAPE;
    const long total = (long)B * 4 * h * w * (C / 4);
    if (total >= (1L << 31)) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(upsample2x_cl_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out,
                       B, h, w, C, mv_fastdiv(C / 4), mv_fastdiv(2 * w), mv_fastdiv(2 * h));
    return mv_check_launch();
}

// gout [B,2h,2w,C] -> gin [B,h,w,C]: the adjoint of the above
extern "C" int mvster_upsample2x_cl_bwd(const float* gout, float* gin, int B, int h, int w, int C, void* stream) {
    if (!gout || !gin) return MVSTER_ERR_NULL;
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 3)) return MVSTER_ERR_SHAPE;
    const long total = (long)B * h * w * (C / 4);
    if (total >= (1L << 31)) return MVSTER_ERR_SHAPE;
    hipLaunchKernelGGL(upsample2x_cl_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gout,
                       gin, B, h, w, C, mv_fastdiv(C / 4), mv_fastdiv(w), mv_fastdiv(h));
    return mv_check_launch();
}
