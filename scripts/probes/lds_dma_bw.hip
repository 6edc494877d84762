// Fetch rate of one CU on gfx950: LDS-DMA (buffer_load_dwordx4 ... lds) against global_load_dwordx4 into registers, by
// footprint of the source (L2-resident and shared by all CUs / per-CU streams from HBM), waves per workgroup and pieces in
// flight per wave.  One workgroup per CU.  Prints aggregate TB/s and bytes per nanosecond and CU.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int DEPTH, bool DMA>
__global__ void __launch_bounds__(512) fetch(const float* src, unsigned bytes, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, (int)bytes, 0x00020000);
    // every (workgroup, wave, iteration, piece) reads its own kilobyte, wrapping around the footprint
    unsigned pos = ((blockIdx.x * nw + wave) * 7919u) * 1024u;
    f32x4v acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const unsigned off = (pos % bytes) + lane * 16;
            pos += 1024u * 131u;
            if (DMA) {
                f32x4v* const dst = reinterpret_cast<f32x4v*>(lds) + (wave * DEPTH + k) * 64;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)dst, 16, off, 0, 0, 0);
            } else {
                const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                acc += v;
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1] + acc[2] + acc[3];
}

template <int DEPTH, bool DMA>
double run(const float* src, unsigned bytes, int waves, int ncu, float* sink) {
    const int iters = 2048 / DEPTH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fetch<DEPTH, DMA>), dim3(ncu), dim3(64 * waves), 64 * 1024, 0, src, bytes, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return (double)ncu * waves * iters * DEPTH * 1024.0 / (ms * 1e-3);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const size_t big = (size_t)1 << 30;
    float *src, *sink;
    hipMalloc(&src, big);
    hipMemset(src, 0, big);
    hipMalloc(&sink, 4096);
    const unsigned foot[3] = {256u << 10, 16u << 20, 1u << 30};
    const char* fname[3] = {"256 KB (L2, shared)", "16 MB", "1 GB (HBM)"};
    printf("%d CUs; bytes/ns/CU = aggregate / CUs (divide by the clock in GHz for bytes per cycle)\n", ncu);
    for (int f = 0; f < 3; ++f)
        for (int waves = 4; waves <= 8; waves += 4) {
            const double a = run<1, true>(src, foot[f], waves, ncu, sink), b = run<4, true>(src, foot[f], waves, ncu, sink),
                         c = run<8, true>(src, foot[f], waves, ncu, sink), d = run<8, false>(src, foot[f], waves, ncu, sink),
                         e = run<16, false>(src, foot[f], waves, ncu, sink);
            printf("%-20s %d waves/CU | LDS-DMA depth 1: %6.2f TB/s (%5.1f B/ns/CU)  depth 4: %6.2f (%5.1f)  depth 8: %6.2f (%5.1f) | "
                   "to registers depth 8: %6.2f (%5.1f)  depth 16: %6.2f (%5.1f)\n",
                   fname[f], waves, a / 1e12, a / 1e9 / ncu, b / 1e12, b / 1e9 / ncu, c / 1e12, c / 1e9 / ncu, d / 1e12, d / 1e9 / ncu,
                   e / 1e12, e / 1e9 / ncu);
        }
    return 0;
}
