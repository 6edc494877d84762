// Argument record and small helpers shared by the convolution kernels (conv_mfma.hip, conv_pers.hip).
#pragma once
#include "common.hpp"

namespace mvconv {

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kMaxTaps = 128;
constexpr int kMaxClasses = 8;

struct ConvArgs {
    const float* in;      // [B, Di, Hi, Wi, CIN]
    const float* wpk;     // packed weights, all classes
    const float* scale;   // [Np]
    const float* shift;   // [Np]
    const float* skip;    // optional
    const float* zeros;   // >= 16 bytes of zeros: the "address" of every padded tap
    const float* prob_w;  // optional fused 1x1x1 head (cout == 8): out becomes [voxels] logits
    const float* prob_b;
    float* out;           // [B, DoF, HoF, WoF, COUT]
    int B, Di, Hi, Wi;
    int Do, Ho, Wo;       // output lattice walked by M (per class)
    int DoF, HoF, WoF;    // full output dims
    int sd, sh, sw;       // input step per lattice step
    int cin;              // input channels (LDS-staged variant; the direct kernel has it as a template argument)
    int cout;             // real output channels
    int ntile_total;      // Np / 16
    int relu;
    int skip_mode;        // 0 none, 1 same-resolution add, 2 bilinear x2 upsample-add (2-D, half resolution)
    int nclass;
    // per class
    int kd[kMaxClasses], kh[kMaxClasses], kw[kMaxClasses];   // sub-kernel extent
    int pd[kMaxClasses], ph[kMaxClasses], pw[kMaxClasses];   // input offset: i = o*s - p + k
    int od[kMaxClasses], oh[kMaxClasses], ow[kMaxClasses];   // output phase
    int osd, osh, osw;                                       // output lattice stride
    int nsteps[kMaxClasses];
    int all_inside[kMaxClasses];                             // no tap of any lattice voxel needs padding
    long woff[kMaxClasses];                                  // float offset of the class's packed weights
    // filled by the C entry point: multiply-shift division by Wo, Ho, Do (valid for dividends < 2^31)
    unsigned div_mul[3], div_shr[3];
    unsigned in_bytes;                                       // size of `in` (< 4 GB): buffer-load range check
};

__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned d, unsigned mul, unsigned shr) {
    return d == 1 ? n : (__umulhi(n, mul) >> shr);
}

static inline void find_divisor(unsigned d, unsigned& mul, unsigned& shr) {
    if (d <= 1) { mul = 0; shr = 0; return; }
    int lg = 31 - __builtin_clz(d);
    if (d & (d - 1)) ++lg;                       // ceil(log2 d)
    const int p = 31 + lg;
    mul = (unsigned)(((1ull << p) + d - 1) / d);
    shr = (unsigned)(p - 32);
}

// conv_pers.hip: persistent LDS-DMA kernel family (variant 5 of mvster_conv_mfma); wpc = workgroups per CU (0 = default)
int dispatch_pers(const ConvArgs& a, int mt, int nt, int wpc, hipStream_t s);
// conv_pers.hip: ping-pong form of the persistent kernel, eight waves per workgroup (variant 7)
int dispatch_pp(const ConvArgs& a, int mt, int nt, hipStream_t s);
// conv_pers.hip: persistent 1x1 kernel with all weights in LDS (variant 6)
int dispatch_1x1(const ConvArgs& a, int mt, int wpc, hipStream_t s);

}  // namespace mvconv
