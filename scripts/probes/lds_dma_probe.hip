// LDS-DMA semantics on gfx950: (1) global_load_lds writes base + lane*16 whatever the per-lane source address;
// (2) buffer_load ... lds with an out-of-range offset writes zeros for that lane (the descriptor range check applies).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip
#include <hip/hip_runtime.h>
typedef float f32x4v __attribute__((ext_vector_type(4)));
__global__ void k(const float* in, float* out, unsigned nbytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // global_load_lds: per-lane pointer, LDS dest = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds(in + (63 - lane) * 4 + wave * 256, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), (short)0, (int)nbytes, 0x00020000);
    unsigned off = (lane & 1) ? 0xFFFFFFF0u : (unsigned)(lane * 16 + wave * 1024);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 1024 + wave * 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = lds[i];
}
int main() {
    float *in, *out; (void)hipMalloc(&in, 4096*4); (void)hipMalloc(&out, 2048*4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i + 1; (void)hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMemset(out, 0xff, 2048 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 2048 * 4, 0, in, out, 4096u * 4u);
    float o[2048]; (void)hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        float want = h[(63 - l) * 4 + w * 256 + j];
        if (o[w * 256 + l * 4 + j] != want) ++bad;
    }
    printf("global_load_lds lane-linear layout: %s (%d bad)\n", bad ? "FAIL" : "ok", bad);
    int badz = 0, badv = 0;
    for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        float got = o[1024 + w * 256 + l * 4 + j];
        if (l & 1) { if (got != 0.0f) ++badz; }
        else if (got != h[l * 4 + w * 256 + j]) ++badv;
    }
    printf("raw_buffer_load_lds: in-range %s (%d bad), out-of-range lanes zero-filled: %s (%d nonzero)\n", badv ? "FAIL" : "ok", badv, badz ? "NO" : "yes", badz);
    return 0;
}
