// Shared declarations for the gfx950 kernels of the MVSTER cost-volume path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
