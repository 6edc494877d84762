// Register budget of a "transform each input slice once" form of the 3x3x3 Winograd layers (compile-only probe; round 6,
// review item 4).  Today (conv_wino.hip: conv_wino_ring_kernel<NCH, 3, ...>) a step = (output slice z, depth tap kz, 16-channel
// chunk): the input slice z + kz - 1 is staged and transformed once per (z, kz), i.e. three times.  The once-only form walks
// the INPUT slices; slice zi feeds output slices zi + 1, zi, zi - 1 through the kd = 0, 1, 2 weights, so three accumulator sets
// (16 transform points x f32x4 each = 64 registers per set and N tile) are live at once and rotate.
//
// This file is the inner step of that form with everything else of the ring kernel stripped to what holds registers: the
// transformed 4x4 block V (16 points x 4 channels = 64 registers), one group of four weight fragments read ahead (2 x 4 x 4 =
// 32), three accumulator sets (192).  Build:  hipcc --offload-arch=gfx950 -O3 -c wino_once_regs.hip
//   -Rpass-analysis=kernel-resource-usage   (scripts/probes/README in DESIGN.md section 8.6 quotes the counts).
#include <hip/hip_runtime.h>

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int SETS>
__global__ void __launch_bounds__(512) wino_once_step(const f32x4v* __restrict__ patch, const f32x4v* __restrict__ u, f32x4v* __restrict__ out,
                                                      int nslices) {
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const ring = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const uring = ring + 4096;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096 + SETS * 1024; i += 512) ring[i] = i < 4096 ? patch[i] : u[i - 4096];
    __syncthreads();
    f32x4v acc[SETS][16];
#pragma unroll
    for (int s = 0; s < SETS; ++s)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[s][q] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    for (int zi = 0; zi < nslices; ++zi) {
        // the slice's 4x4 block of this lane and its transform V = B^T d B (as in conv_wino.hip: 16 ds_read_b128, 64 packed adds)
        f32x4v d[4][4], V[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int x = 0; x < 4; ++x) d[r][x] = ring[((zi & 3) * 16 + r * 4 + x) * 64 + lane];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const f32x4v t0 = d[0][x] - d[2][x], t1 = d[1][x] + d[2][x], t2 = d[2][x] - d[1][x], t3 = d[1][x] - d[3][x];
            d[0][x] = t0; d[1][x] = t1; d[2][x] = t2; d[3][x] = t3;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            V[r][0] = d[r][0] - d[r][2];
            V[r][1] = d[r][1] + d[r][2];
            V[r][2] = d[r][2] - d[r][1];
            V[r][3] = d[r][1] - d[r][3];
        }
        // one transform, SETS depth taps: accumulator set s takes the weights of tap s
#pragma unroll
        for (int s = 0; s < SETS; ++s) {
            f32x4v ub[2][4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) ub[0][qq] = uring[(s * 16 + qq) * 64 + lane];
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                if (grp < 3) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) ub[(grp + 1) & 1][qq] = uring[(s * 16 + (grp + 1) * 4 + qq) * 64 + lane];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
                        acc[s][grp * 4 + qq] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[grp & 1][qq][j], V[grp][qq][j], acc[s][grp * 4 + qq], 0, 0, 0);
            }
        }
        // the set whose third contribution has landed leaves (output transform A^T M A), the sets rotate
        {
            f32x4v t[2][4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                t[0][x] = acc[0][0 + x] + acc[0][4 + x] + acc[0][8 + x];
                t[1][x] = acc[0][4 + x] - (acc[0][8 + x] + acc[0][12 + x]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                out[((zi * 2 + i) * 2 + 0) * 512 + threadIdx.x] = t[i][0] + t[i][1] + t[i][2];
                out[((zi * 2 + i) * 2 + 1) * 512 + threadIdx.x] = t[i][1] - (t[i][2] + t[i][3]);
            }
        }
#pragma unroll
        for (int s = 0; s + 1 < SETS; ++s)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[s][q] = acc[s + 1][q];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[SETS - 1][q] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    }
}

template __global__ void wino_once_step<1>(const f32x4v*, const f32x4v*, f32x4v*, int);
template __global__ void wino_once_step<3>(const f32x4v*, const f32x4v*, f32x4v*, int);
