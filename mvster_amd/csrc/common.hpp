// Shared declarations for the gfx950 kernels of the MVSTER cost-volume path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "mvster_math.h"

// error codes of the C ABI (include/mvster_hip.h)
#define MVSTER_OK 0
#define MVSTER_ERR_NULL -1
#define MVSTER_ERR_SHAPE -2
#define MVSTER_ERR_UNSUPPORTED -3
#define MVSTER_ERR_LAUNCH -4

static inline int mv_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MVSTER_OK : MVSTER_ERR_LAUNCH;
}

// What the last launcher called on this host thread dispatched, spelled like the profiler spells the kernel
// ("conv_lds_kernel<2, 1, 3, 1, 3>"): mvster_last_kernel() hands it to the caller, so that bench.py attributes its
// event timings to the kernel the library really chose instead of re-deriving the dispatch rules.
extern thread_local const char* mv_last_kernel;
struct MvKernelName {          // formatted once per call site; function-local statics initialise thread-safely (C++11)
    char s[96];
    template <class... A>
    explicit MvKernelName(const char* fmt, A... a) {
        if constexpr (sizeof...(A) == 0) snprintf(s, sizeof s, "%s", fmt);
        else snprintf(s, sizeof s, fmt, a...);
    }
};
#define MV_NOTE_KERNEL(...)                              \
    do {                                                 \
        static const MvKernelName nm_(__VA_ARGS__);      \
        mv_last_kernel = nm_.s;                          \
    } while (0)

// Probe build (make -C mvster_amd/csrc probes: -DMVSTER_PROBES, libmvster_hip_probes.so).  The product library reads NO
// environment variable and carries none of the measured-but-not-chosen kernel forms: experiment switches resolve to "unset"
// here, and the kernels kept for the record (pixel-major and LDS-window warp kernels, the ping-pong convolution) compile
// only into the probe library, which the probe scripts and the probe-marked GPU tests load through MVSTER_LIB.
#ifdef MVSTER_PROBES
#define MV_PROBE_ENV(name) getenv(name)
#else
#define MV_PROBE_ENV(name) (static_cast<const char*>(nullptr))
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Cache policy of the persistent kernels' output stores (aux operand of the buffer-store builtins; 16 = sc1 = write-through).
// A plain store leaves its line dirty in the XCD's L2, and a kernel that writes tens of MB ends with up to 32 MB of dirty
// lines whose write-back the next kernel's start waits for.  Written through, the data leaves during the kernel: ISOLATED
// (the same launch back to back) the 8 -> 8 narrow layer at 5 x 512 x 640 runs 26.3 instead of 29.7 us, 4 -> 8 17.3 instead of
// 19.7 (profiles/r04_a_conv_narrow_writethrough.txt).  INSIDE the forward it is neutral: 1 092 / 1 094 against 1 093 / 1 088
// depth-maps/s with two depth maps in flight, one forward alone 1.142 against 1.150 ms (same box, alternating runs,
// profiles/r04_a_bench_writethrough_ab.txt) -- the other depth map's kernels fill the write-back window, and a consumer
// whose tiles xcd_remap places on the producer's XCD finds the tail of the producer's output in that L2, which a
// write-through store drops.  Plain stores stay the default; -DMV_STORE_AUX=16 builds the other form.
#ifndef MV_STORE_AUX
#define MV_STORE_AUX 0
#endif

// XCD-aware workgroup order.  MI355X dispatches workgroup b to XCD b % 8 and every XCD has its own 4 MB
// L2, so spatially adjacent tiles (which share halo rows, source texels and weights) would land on eight
// different L2s.  This bijection hands each XCD a contiguous 1/8 of the tile range instead; it only ever
// changes speed (placement is not a correctness contract).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
