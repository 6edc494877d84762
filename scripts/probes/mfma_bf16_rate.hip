// Issue rate of the bf16 MFMAs on gfx950, one wave per SIMD and two: cycles per instruction for streams of independent
// v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 (s_memtime around 4 x 256 instructions).
// hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_rate_probe mfma_bf16_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(const bf16x8* a, float* out, unsigned long long* t) {
    bf16x8 A = a[threadIdx.x & 63], B = a[64 + (threadIdx.x & 63)];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 256 / NACC * 4; ++r) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC>
__global__ void k32(const bf16x8* a, float* out, unsigned long long* t) {
    bf16x8 A = a[threadIdx.x & 63], B = a[64 + (threadIdx.x & 63)];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < 256 / NACC * 4; ++r) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main() {
    bf16x8* a; float* out; unsigned long long* t;
    hipMalloc(&a, 128 * 16); hipMemset(a, 0, 128 * 16);
    hipMalloc(&out, 1 << 22); hipMalloc(&t, 1 << 16);
    unsigned long long h[64];
#define RUN(K, name, threads)                                                                   \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(K, dim3(256), dim3(threads), 0, 0, a, out, t); \
    hipDeviceSynchronize(); hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);                   \
    printf("%-44s %3d threads/wg: %.1f cycles per MFMA per wave\n", name, threads, (double)h[0] / 1024.0);
    RUN(k16<1>, "16x16x32 bf16, 1 accumulator (dependent)", 256)
    RUN(k16<2>, "16x16x32 bf16, 2 accumulators", 256)
    RUN(k16<4>, "16x16x32 bf16, 4 accumulators", 256)
    RUN(k16<8>, "16x16x32 bf16, 8 accumulators", 256)
    RUN(k16<8>, "16x16x32 bf16, 8 accumulators", 512)
    RUN(k32<1>, "32x32x16 bf16, 1 accumulator (dependent)", 256)
    RUN(k32<2>, "32x32x16 bf16, 2 accumulators", 256)
    RUN(k32<4>, "32x32x16 bf16, 4 accumulators", 256)
    RUN(k32<4>, "32x32x16 bf16, 4 accumulators", 512)
    return 0;
}
