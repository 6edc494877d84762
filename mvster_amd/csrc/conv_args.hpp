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

// ---- shared by the persistent LDS-DMA kernels (conv_pers.hip, conv_wino.hip) ------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct PersArgs {
    unsigned tiles_x, tiles_y, ntiles;       // tiles per (b, z) slice and in total
    unsigned out_bytes;                      // size of `out` (and of a same-shape skip): < 2^31
    int prio;                                // 1: waves in odd slots of their SIMD run at raised priority (see kernel)
    unsigned mul[3], shr[3], one[3];         // multiply-shift division by tiles_x, tiles_y, Do; one = ~0 if the divisor is 1
};

struct TilePos { int b, zo, ty0, tx0; };

template <int MT, int KW, int SW, int KD>
struct PersGeom {
    static constexpr int TY = 2 * MT;
    static constexpr int KH = KW;
    static constexpr int PW = 31 * SW + KW;                  // patch width (input pixels)
    static constexpr int PH = (TY - 1) * SW + KH;            // patch height of one depth slice
    static constexpr int ROWS = KD * PH;
    static constexpr int PWH = (PW + 1) / 2;                 // stride 2: columns per parity
    static constexpr int ROWSLOTS = SW == 1 ? PW * 2 : PWH * 4;           // float4 slots of one patch row in one plane
    static constexpr int USED = ROWS * ROWSLOTS;                          // slots of one plane that hold pixels
    static constexpr int NBLK = (USED + 63) / 64;                         // DMA wave-instructions per plane
    static constexpr int PLANE = ((NBLK * 64 + 7) & ~7) + 4;              // plane pitch (float4), = 4 mod 8
};

// host side: tile counts, sizes and the multiply-shift divisors of a launch whose workgroup tile is TY x 32 output pixels
// of one (b, z) slice.  false: a tensor of 2 GB or more (32-bit byte offsets; 0x80000000 + any offset must stay out of range)
static inline bool fill_pers_args(const ConvArgs& a, int TY, PersArgs& p) {
    p.tiles_x = (unsigned)((a.Wo + 31) / 32);
    p.tiles_y = (unsigned)((a.Ho + TY - 1) / TY);
    const long ntiles = (long)p.tiles_x * p.tiles_y * a.Do * a.B;
    const long out_bytes = (long)a.B * a.DoF * a.HoF * a.WoF * a.cout * 4;
    if (ntiles >= (1L << 30) || a.in_bytes >= (1u << 31) || out_bytes >= (1L << 31)) return false;
    p.ntiles = (unsigned)ntiles;
    p.out_bytes = (unsigned)out_bytes;
    p.prio = 0;
    const unsigned divisors[3] = {p.tiles_x, p.tiles_y, (unsigned)a.Do};
    for (int i = 0; i < 3; ++i) {
        find_divisor(divisors[i], p.mul[i], p.shr[i]);
        p.one[i] = divisors[i] == 1 ? ~0u : 0u;
    }
    return true;
}

// compute units of the current device (0 on error); conv_pers.hip
int num_cus();

// Dynamic LDS above 64 KB has to be allowed per kernel AND per device (a process may drive several GPUs: nn.DataParallel
// replicas, one thread per device): one bit per device ordinal in the caller's static mask.
static inline bool allow_big_lds(const void* kern, unsigned long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1ul)) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    if (dev >= 0 && dev < 64) done |= 1ul << dev;
    return true;
}

// conv_pers.hip: persistent LDS-DMA kernel family (variant 5 of mvster_conv_mfma); wpc = workgroups per CU (0 = default)
int dispatch_pers(const ConvArgs& a, int mt, int nt, int wpc, hipStream_t s);
// conv_pers.hip: ping-pong form of the persistent kernel, eight waves per workgroup (variant 7)
int dispatch_pp(const ConvArgs& a, int mt, int nt, hipStream_t s);
// conv_wino.hip: Winograd F(2x2, 3x3) forms of the persistent kernel (variants 8 and 9 = ring; `wpk` = the transformed weights)
int dispatch_wino(const ConvArgs& a, int nt, int wpc, bool ring, hipStream_t s);
// conv_b3.hip: fp32 products as six bf16 MFMAs (3-way split operands), 3x3 / 3x3x3 stride-1 layers (variant 11; `wpk` = the
// pre-split bf16 weight fragments)
int dispatch_b3(const ConvArgs& a, int mt, int wpc, hipStream_t s);
// conv_pers.hip: persistent 1x1 kernel with all weights in LDS (variant 6)
int dispatch_1x1(const ConvArgs& a, int mt, int wpc, hipStream_t s);

// deconv_select.hip: reg2d's last layer + `prob` + the depth selection on MFMA tiles, persistent LDS-DMA ring (D in {4, 8};
// MVSTER_ERR_UNSUPPORTED otherwise: the VALU kernel of conv_small.hip runs)
int dispatch_deconv_select_mfma(const float* in, const float* w, const float* scale, const float* shift, const float* skip,
                                const float* prob_w, const float* prob_b, const float* hypo, float* attn, float* depth,
                                float* conf, float* inv_min, float* inv_max, float* logits_out, int B, int D, int Hi, int Wi,
                                int relu, float split_itv, hipStream_t s);

}  // namespace mvconv
