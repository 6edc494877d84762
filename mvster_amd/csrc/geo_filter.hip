// Geometric-consistency filter of the depth-map fusion step: what the reference does per
// (reference view, source view) pair with NumPy + cv2.remap on the CPU (test_mvs4.py:273-328) and then sums over
// the source views (:362-385), as ONE launch per reference view.  One thread per reference pixel walks all source
// views: lift the pixel with the reference depth, project into the source view, sample the source depth map
// (cv2.remap INTER_LINEAR semantics: coordinates quantised to 1/32 pixel, constant-0 border), lift with the
// sampled depth, project back, and keep the view if the pixel lands within 1 px and 1 % relative depth.
// HBM-bound and tiny: per pixel and view 4 gathered floats; the geometry runs in fp64 like NumPy's
// (float32 matrices x float64 points), the maps and the comparisons in fp32 exactly where the reference casts.
#include "common.hpp"

namespace {

constexpr int kViewDoubles = 42;   // per source view: A[3x4] = E_src inv(E_ref), K_src[3x3], inv(K_src)[3x3], B[3x4] = E_ref inv(E_src)

struct GeoArgs {
    const float* depth_ref;    // [H, W]
    const float* depth_src;    // [NS, H, W]
    const double* ref_mats;    // inv(K_ref)[9], K_ref[9]
    const double* view_mats;   // [NS, 42]
    int* mask_sum;             // [H, W]
    float* depth_sum;          // [H, W]  sum over views of the masked reprojected depth (view order)
    unsigned char* view_mask;  // optional [NS, H, W]
    float* view_depth;         // optional [NS, H, W]
    float* x_src;              // optional [NS, H, W]
    float* y_src;              // optional [NS, H, W]
    int NS, H, W;
    float pix_thres, rel_thres;
};

__device__ __forceinline__ void mat3(const double* m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = fma(m[2], z, fma(m[1], y, m[0] * x));
    oy = fma(m[5], z, fma(m[4], y, m[3] * x));
    oz = fma(m[8], z, fma(m[7], y, m[6] * x));
}

__device__ __forceinline__ void mat34(const double* m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = fma(m[2], z, fma(m[1], y, m[0] * x)) + m[3];
    oy = fma(m[6], z, fma(m[5], y, m[4] * x)) + m[7];
    oz = fma(m[10], z, fma(m[9], y, m[8] * x)) + m[11];
}

// cv2.remap(src, x, y, INTER_LINEAR), BORDER_CONSTANT 0: fixed-point coordinates with 5 fractional bits
__device__ __forceinline__ float remap_linear(const float* __restrict__ src, int H, int W, float x, float y) {
    if (!(fabsf(x) < 1e6f) || !(fabsf(y) < 1e6f)) return 0.0f;        // also NaN: every tap is outside
    const int sx = (int)rintf(x * 32.0f), sy = (int)rintf(y * 32.0f);  // cvRound (round half to even)
    const int ix = sx >> 5, iy = sy >> 5;
    const float fx = (float)(sx & 31) * (1.0f / 32.0f), fy = (float)(sy & 31) * (1.0f / 32.0f);
    const bool x0 = (unsigned)ix < (unsigned)W, x1 = (unsigned)(ix + 1) < (unsigned)W;
    const bool y0 = (unsigned)iy < (unsigned)H, y1 = (unsigned)(iy + 1) < (unsigned)H;
    const int cx0 = min(max(ix, 0), W - 1), cx1 = min(max(ix + 1, 0), W - 1);
    const int cy0 = min(max(iy, 0), H - 1), cy1 = min(max(iy + 1, 0), H - 1);
    const float v00 = src[(long)cy0 * W + cx0], v01 = src[(long)cy0 * W + cx1];
    const float v10 = src[(long)cy1 * W + cx0], v11 = src[(long)cy1 * W + cx1];
    const float s00 = (y0 && x0) ? v00 : 0.0f, s01 = (y0 && x1) ? v01 : 0.0f;
    const float s10 = (y1 && x0) ? v10 : 0.0f, s11 = (y1 && x1) ? v11 : 0.0f;
    const float w00 = mv::mul_rn(1.0f - fy, 1.0f - fx), w01 = mv::mul_rn(1.0f - fy, fx);
    const float w10 = mv::mul_rn(fy, 1.0f - fx), w11 = mv::mul_rn(fy, fx);
    return mv::add_rn(mv::add_rn(mv::add_rn(mv::mul_rn(s00, w00), mv::mul_rn(s01, w01)), mv::mul_rn(s10, w10)),
                      mv::mul_rn(s11, w11));
}

__global__ void __launch_bounds__(256) geo_filter_kernel(GeoArgs a) {
    const long hw = (long)a.H * a.W;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int y = (int)(p / a.W), x = (int)(p - (long)y * a.W);
    const float dref = a.depth_ref[p];
    const double d = (double)dref;
    double rx, ry, rz;
    mat3(a.ref_mats, (double)x * d, (double)y * d, d, rx, ry, rz);     // inv(K_ref) @ ((x, y, 1) * depth)
    int count = 0;
    float dsum = 0.0f;
    for (int v = 0; v < a.NS; ++v) {
        const double* m = a.view_mats + (long)v * kViewDoubles;
        double qx, qy, qz, kx, ky, kz;
        mat34(m, rx, ry, rz, qx, qy, qz);                              // source camera space
        mat3(m + 12, qx, qy, qz, kx, ky, kz);                          // K_src @ .
        const double xs = kx / kz, ys = ky / kz;
        const float xsf = (float)xs, ysf = (float)ys;
        const float sampled = remap_linear(a.depth_src + (long)v * hw, a.H, a.W, xsf, ysf);
        const double sd = (double)sampled;
        double sx3, sy3, sz3, bx, by, bz, ux, uy, uz;
        mat3(m + 21, xs * sd, ys * sd, sd, sx3, sy3, sz3);             // inv(K_src) @ ((xs, ys, 1) * sampled)
        mat34(m + 30, sx3, sy3, sz3, bx, by, bz);                      // back in the reference camera space
        float drep = (float)bz;
        mat3(a.ref_mats + 9, bx, by, bz, ux, uy, uz);                  // K_ref @ .
        const float xr = (float)(ux / uz), yr = (float)(uy / uz);
        const double ex = (double)xr - (double)x, ey = (double)yr - (double)y;
        const double dist = sqrt(ex * ex + ey * ey);
        const float rel = mv::div_rn(fabsf(mv::sub_rn(drep, dref)), dref);
        const bool ok = dist < (double)a.pix_thres && rel < a.rel_thres;
        if (!ok) drep = 0.0f;
        count += ok ? 1 : 0;
        dsum = mv::add_rn(dsum, drep);
        if (a.view_mask) a.view_mask[(long)v * hw + p] = ok ? 1 : 0;
        if (a.view_depth) a.view_depth[(long)v * hw + p] = drep;
        if (a.x_src) a.x_src[(long)v * hw + p] = xsf;
        if (a.y_src) a.y_src[(long)v * hw + p] = ysf;
    }
    a.mask_sum[p] = count;
    a.depth_sum[p] = dsum;
}

}  // namespace

extern "C" int mvster_geo_filter(const float* depth_ref, const float* depth_src, const double* ref_mats,
                                 const double* view_mats, int* mask_sum, float* depth_sum, unsigned char* view_mask,
                                 float* view_depth, float* x_src, float* y_src, int NS, int H, int W, float pix_thres,
                                 float rel_thres, void* stream) {
    if (!depth_ref || !depth_src || !ref_mats || !view_mats || !mask_sum || !depth_sum) return MVSTER_ERR_NULL;
    if (NS <= 0 || H <= 0 || W <= 0) return MVSTER_ERR_SHAPE;
    GeoArgs a;
    a.depth_ref = depth_ref; a.depth_src = depth_src; a.ref_mats = ref_mats; a.view_mats = view_mats;
    a.mask_sum = mask_sum; a.depth_sum = depth_sum; a.view_mask = view_mask; a.view_depth = view_depth;
    a.x_src = x_src; a.y_src = y_src; a.NS = NS; a.H = H; a.W = W; a.pix_thres = pix_thres; a.rel_thres = rel_thres;
    const long hw = (long)H * W;
    hipLaunchKernelGGL(geo_filter_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}
