// Cost of LDS atomics on gfx950: wave64 ds_add_f32 / ds_add_u32 / ds_add_u64 / plain read-modify-write, distinct addresses per lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters) {
    __shared__ float buf[4096];
    __shared__ unsigned long long buf64[2048];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0.f;
    for (int i = threadIdx.x; i < 2048; i += 256) buf64[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v = 1.0f + lane * 0.001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int idx = ((it * 8 + k) * 64 + lane + wave * 17) & 4095;   // distinct banks per lane, waves overlap
            if (MODE == 0) unsafeAtomicAdd(&buf[idx], v);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&buf[idx]), (unsigned)lane);
            else if (MODE == 2) atomicAdd(&buf64[idx & 2047], (unsigned long long)lane);
            else if (MODE == 3) buf[idx] = buf[idx] + v;                       // plain RMW (racy across waves: timing only)
            else if (MODE == 4) atomicAdd(&buf[idx], v);                       // safe float atomic
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = buf[1] + (float)buf64[1];
}

int main() {
    float* out; hipMalloc(&out, 4096 * 4);
    const char* names[] = {"ds_add_f32 (unsafeAtomicAdd)", "ds_add_u32", "ds_add_u64", "plain read+add+write", "atomicAdd(float)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 2;
    for (int m = 0; m < 5; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
            if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
            if (m == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters);
            if (m == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, out, iters);
            if (m == 4) hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 2 blocks x 4 waves x iters x 8 wave-instructions
        const double instr_per_cu = 2.0 * 4 * iters * 8;
        printf("%-32s %8.3f ms  -> %7.1f cycles per wave64 instruction per CU (2.4 GHz)\n", names[m], ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
    }
    return 0;
}
