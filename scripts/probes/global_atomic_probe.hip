// Throughput of global fp32 atomic adds on gfx950 to scattered texels of 8 consecutive floats (the out-of-window taps of
// the warp/aggregation backward), by how the 8 channels of a texel are issued:
//   0: lane = texel, 8 successive instructions walk the channels (what a pixel-per-lane kernel does naturally)
//   1: lane = (texel, channel): 8 neighbouring lanes cover one texel's 32 contiguous bytes, one instruction
//   2: as 1, 16 consecutive floats per texel (64 bytes)
// Texel addresses: pseudo-random over a 10 x 512 x 640 x 8 map (210 MB) or over a 2 MB region (L2 resident).
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* buf, unsigned ntexel, int iters) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            const unsigned t = hash(gid * 7919u + it) % ntexel;
#pragma unroll
            for (int c = 0; c < 8; ++c) unsafeAtomicAdd(buf + (size_t)t * 8 + c, 1.0f);
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned t = hash((gid >> 3) * 7919u + it * 8 + k) % ntexel;
                unsafeAtomicAdd(buf + (size_t)t * 8 + (gid & 7), 1.0f);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned t = hash((gid >> 4) * 7919u + it * 8 + k) % (ntexel / 2);
                unsafeAtomicAdd(buf + (size_t)t * 16 + (gid & 15), 1.0f);
            }
        }
    }
}

int main() {
    const size_t ntexel_big = (size_t)10 * 512 * 640;
    float* buf; hipMalloc(&buf, ntexel_big * 8 * 4); hipMemset(buf, 0, ntexel_big * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 64, blocks = 256 * 16;
    const char* names[] = {"lane = texel, channels in 8 instructions", "lane = (texel, channel), 32 B runs", "lane = (texel, channel), 64 B runs"};
    for (int region = 0; region < 2; ++region) {
        const unsigned ntexel = region == 0 ? (unsigned)ntexel_big : 65536u;
        for (int m = 0; m < 3; ++m) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (m == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, buf, ntexel, iters);
                if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, buf, ntexel, iters);
                if (m == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, buf, ntexel, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double adds = (double)blocks * 256 * iters * 8;
            printf("%-8s %-44s %8.3f ms  %7.1f G float-adds/s\n", region == 0 ? "210 MB" : "2 MB", names[m], ms, adds / ms / 1e6);
        }
    }
    return 0;
}
