// Channels-last implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
// One kernel family covers every dense contraction of the path (eval mode, BatchNorm
// folded into a per-channel scale/shift):
//   reg2d / reg3d   Conv3d (1,3,3) s1/s2, 3x3x3, ConvTranspose3d (1,3,3)|3x3x3 s2
//                   models/mvs4net_utils.py:870-965 (+ ConvBnReLU3D :116-123)
//   FPN4            Conv2d 3x3, 5x5 s2, 1x1 (+bias), bilinear x2 upsample-add
//                   models/mvs4net_utils.py:419-502 (+ Conv2d :224-251)
//
// GEMM view:  M = output voxels (b,z,y,x flattened), N = Cout, K = taps x Cin.
//   D[m][n] += sum_k A[m][k] * B[k][n]   with v_mfma_f32_16x16x4_f32 (exact fp32, k-ordered FMA chain)
//   A[m][k] : lane (m = lane&15, q = lane>>4) reads ONE float4 = 4 consecutive input channels of its
//             voxel for the tap that K-slot belongs to -> feeds 4 MFMAs (the K order inside a 16-wide
//             K step is a fixed permutation shared with the packed weights, so it cancels out)
//   B[k][n] : weights pre-packed on the host in exactly the fragment order -> one coalesced float4 per lane
//   D       : the two operands are passed SWAPPED (weights in the A slot), so the accumulator is D^T:
//             lane holds output channels 4*(lane>>4)+r (r = 0..3) of voxel lane&15 -> float4 stores
// A wave owns MT x NT tiles of 16x16; there is no LDS staging and no barrier in the K loop: the
// activations of one stage are L2/MALL resident (<= 42 MB), every tap re-read is a cache hit, and
// zero padding is a per-lane select.  Transposed stride-2 layers run as 2^k parity classes
// (blockIdx.z), each an ordinary small-kernel convolution on the input lattice.
//
// Epilogue (fused): y = acc*scale[n] + shift[n]; optional ReLU; optional skip add (same-resolution
// tensor, or bilinear x2 align_corners upsample of a half-resolution tensor); channels-last store.
//
// Roofline: the 3x3x3 layers are MFMA-bound (157.3 TFLOP/s fp32 matrix peak), the full-resolution
// (1,3,3) layers are HBM/L2-bound.  FLOPs per launch = 2 * M * Cout * taps * Cin.
#include <stdlib.h>

#include "conv_args.hpp"

namespace {

using namespace mvconv;

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// Probe build only (make timeline -> libmvster_hip_tl.so, scripts/conv_timeline.py): lane 0 of every wavefront stamps
// s_memtime at the kernel's phase boundaries into a debug buffer.  The product library is compiled without it.
#ifdef MVSTER_TIMELINE
__device__ unsigned long long* g_tl = nullptr;
#define MV_TL(k)                                                                                              \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (g_tl && (threadIdx.x & 63) == 0)                                                                  \
            g_tl[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime();     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#define MV_TL_ID()                                                                                            \
    do {                                                                                                      \
        if (g_tl && (threadIdx.x & 63) == 0) {                                                                \
            g_tl[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 6] = __builtin_amdgcn_s_getreg(63492);   \
            g_tl[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + 7] = __builtin_amdgcn_s_getreg(63508);   \
        }                                                                                                     \
    } while (0)
#else
#define MV_TL(k)
#define MV_TL_ID()
#endif

// Fused epilogue for 4 consecutive output channels n0..n0+3 of one output voxel.
// (scale / shift are padded to 16 * ntiles floats: one 16-byte load each; callers that can, load them ahead of the MFMAs)
__device__ __forceinline__ void epilogue_store(const ConvArgs& a, f32x4v v, int n0, long opix, int b, int oy, int ox,
                                               const f32x4v sc, const f32x4v sh) {
    const bool vec = (n0 + 3 < a.cout) && ((a.cout & 3) == 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = fmaf(v[j], sc[j], sh[j]);
        if (a.relu) v[j] = fmaxf(v[j], 0.0f);
    }
    if (a.prob_w) {
        // fused `prob` head (reference reg2d.prob, mvs4net_utils.py:900): 8-channel dot product; the two
        // lanes holding channels 0-3 and 4-7 of this voxel are 16 apart in the wave
        if (a.skip_mode == 1) {
            const f32x4v k = *reinterpret_cast<const f32x4v*>(a.skip + opix * a.cout + n0);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += k[j];
        }
        float part = v[0] * a.prob_w[n0];
#pragma unroll
        for (int j = 1; j < 4; ++j) part = fmaf(v[j], a.prob_w[n0 + j], part);
        const float other = __shfl_xor(part, 16);
        if (n0 == 0) a.out[opix] = (part + other) + a.prob_b[0];
        return;
    }
    float* op = a.out + opix * a.cout + n0;
    if (vec) {
        if (a.skip_mode == 1) {
            const f32x4v k = *reinterpret_cast<const f32x4v*>(a.skip + opix * a.cout + n0);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += k[j];
        } else if (a.skip_mode == 2) {
            // F.interpolate(prev, x2, bilinear, align_corners) + inner(conv)   (mvs4net_utils.py:482-488)
            const int hh = a.HoF / 2, wh = a.WoF / 2;
            const mv::Lerp ly = mv::make_lerp(oy, hh, a.HoF), lx = mv::make_lerp(ox, wh, a.WoF);
            const float* sk = a.skip + (long)b * hh * wh * a.cout + n0;
            const f32x4v k00 = *reinterpret_cast<const f32x4v*>(sk + ((long)ly.i0 * wh + lx.i0) * a.cout);
            const f32x4v k01 = *reinterpret_cast<const f32x4v*>(sk + ((long)ly.i0 * wh + lx.i1) * a.cout);
            const f32x4v k10 = *reinterpret_cast<const f32x4v*>(sk + ((long)ly.i1 * wh + lx.i0) * a.cout);
            const f32x4v k11 = *reinterpret_cast<const f32x4v*>(sk + ((long)ly.i1 * wh + lx.i1) * a.cout);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = mv::bilerp(ly, lx, k00[j], k01[j], k10[j], k11[j]) + v[j];
        }
        *reinterpret_cast<f32x4v*>(op) = v;
    } else {
        // channel count not a multiple of 4 (e.g. the 1-channel prob head of reg3d)
        for (int j = 0; j < 4; ++j) {
            if (n0 + j >= a.cout) break;
            float r = v[j];
            if (a.skip_mode == 1) r += a.skip[opix * a.cout + n0 + j];
            op[j] = r;
        }
    }
}

__device__ __forceinline__ void epilogue_store(const ConvArgs& a, f32x4v v, int n0, long opix, int b, int oy, int ox) {
    epilogue_store(a, v, n0, opix, b, oy, ox, *reinterpret_cast<const f32x4v*>(a.scale + n0),
                   *reinterpret_cast<const f32x4v*>(a.shift + n0));
}

// SPLITK: the 4 waves of a workgroup share ONE set of MT x NT tiles and each takes every 4th K step; the
// partial accumulators are summed through LDS in a fixed order (deterministic).  For the small, deep
// layers (e.g. 8x10x8 voxels x 64 channels, K = 1728) this turns one 108-step dependent chain per wave
// into four 27-step chains and quadruples the number of resident waves.
// Occupancy target per tile shape (waves per SIMD = workgroups of 4 waves per CU): the direct kernel hides its
// operand latency with co-resident waves, so the register allocator is told to fit one more wave than it would
// pick on its own for the register-heavy tiles (measured per layer by the tuner).
constexpr int conv_min_waves(int mt, int nt) {
    const int t = mt * nt;
    return t <= 2 ? 5 : (t <= 4 ? 4 : (t <= 8 ? 3 : 2));     // 2x5 at 3 waves spills (77 -> 79 us), left at 2
}

template <int CIN, int MT, int NT, bool SPLITK>
__global__ void __launch_bounds__(256, conv_min_waves(MT, NT)) conv_mfma_kernel(ConvArgs a) {
    static_assert(CIN % 4 == 0 && (CIN >= 16 ? CIN % 16 == 0 : 16 % CIN == 0), "channel packing");
    __shared__ f32x4v red[SPLITK ? 3 * MT * NT * 64 : 1];
    __shared__ int lut_ofs[kMaxTaps];   // linear input offset (voxels) of a tap
    __shared__ int lut_zyx[kMaxTaps];   // kz | ky << 8 | kx << 16

    const int cls = blockIdx.z;
    const int KD = a.kd[cls], KH = a.kh[cls], KW = a.kw[cls];
    const int ntaps = KD * KH * KW;
    for (int t = threadIdx.x; t < ntaps; t += blockDim.x) {
        const int kx = t % KW, ky = (t / KW) % KH, kz = t / (KW * KH);
        lut_ofs[t] = (kz * a.Hi + ky) * a.Wi + kx;
        lut_zyx[t] = kz | (ky << 8) | (kx << 16);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lm = lane & 15;   // A row / B,D column
    const int lq = lane >> 4;   // K slot
    // all voxel counts fit 32 bits (checked on the host side of the ABI)
    const unsigned Mtot = (unsigned)(a.B * a.Do * a.Ho * a.Wo);
    const int nt0 = blockIdx.y * NT;
    const unsigned ntiles = (Mtot + 15u) >> 4;
    // Grid-stride over tile groups (the host launches one workgroup per group; see launch()).
    const unsigned ngroups = SPLITK ? (ntiles + MT - 1) / MT : (ntiles + 4 * MT - 1) / (4 * MT);
    for (unsigned grp0 = blockIdx.x; grp0 < ngroups; grp0 += gridDim.x) {
    const unsigned grp = gridDim.x == ngroups ? xcd_remap(grp0, ngroups) : grp0;
    const unsigned tile0 = SPLITK ? grp * MT : (grp * 4u + wave) * MT;   // first 16-voxel tile
    if (!SPLITK && tile0 * 16u >= Mtot) break;

    // A-role voxel of this lane in each M tile.  The same voxel is the one this lane stores in the
    // epilogue (swapped MFMA operands), so its output index is computed once, here.
    int iz0[MT], iy0[MT], ix0[MT];
    int pin0[MT];
    int opix[MT];      // output voxel index, -1 = none
    int oyx[MT];       // oy << 16 | ox (for the upsample-add epilogue), b is opix / (DoF*HoF*WoF)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        unsigned m = (tile0 + mt) * 16u + lm;
        const bool ok = m < Mtot;
        if (!ok) m = 0;
        unsigned r = fast_div(m, a.Wo, a.div_mul[0], a.div_shr[0]);
        const int x = (int)(m - r * (unsigned)a.Wo);
        unsigned r2 = fast_div(r, a.Ho, a.div_mul[1], a.div_shr[1]);
        const int y = (int)(r - r2 * (unsigned)a.Ho);
        const unsigned b = fast_div(r2, a.Do, a.div_mul[2], a.div_shr[2]);
        const int z = (int)(r2 - b * (unsigned)a.Do);
        iz0[mt] = ok ? z * a.sd - a.pd[cls] : -(1 << 20);
        iy0[mt] = y * a.sh - a.ph[cls];
        ix0[mt] = x * a.sw - a.pw[cls];
        pin0[mt] = (((int)b * a.Di + (ok ? iz0[mt] : 0)) * a.Hi + iy0[mt]) * a.Wi + ix0[mt];
        const int oz = z * a.osd + a.od[cls], oy = y * a.osh + a.oh[cls], ox = x * a.osw + a.ow[cls];
        opix[mt] = ok ? (((int)b * a.DoF + oz) * a.HoF + oy) * a.WoF + ox : -1;
        oyx[mt] = (oy << 16) | ox;
    }

    f32x4v acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    const float* wp = a.wpk + a.woff[cls] + ((long)nt0 * 64 + lane) * 4;
    const long wstep = (long)a.ntile_total * 256;
    const int nsteps_all = a.nsteps[cls];
    // K steps of this wave: first, first+stride, ... (all of them unless SPLITK)
    const int kfirst = SPLITK ? wave : 0, kstride = SPLITK ? 4 : 1;
    const int nsteps = SPLITK ? (nsteps_all - wave + 3) / 4 : nsteps_all;

    // Activations are read through a buffer descriptor: the offset of a padded tap is pushed past the
    // end of the tensor and the hardware range check returns zeros -- no branch, no select on the data,
    // 32-bit address arithmetic only.
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
    const bool no_pad = a.all_inside[cls] != 0;   // 1x1-style classes: every tap of every voxel is inside
    f32x4v af[MT], bf[NT];
    auto load_step = [&](int si, f32x4v (&A)[MT], f32x4v (&Bv)[NT]) {
        const int s = kfirst + si * kstride;
        const int kk = s * 16 + lq * 4;
        const int tap = kk / CIN;
        const int c = kk % CIN;
        const bool tap_ok = tap < ntaps;
        const int tt = tap_ok ? tap : 0;
        const int ofs = lut_ofs[tt];
        const int zyx = lut_zyx[tt];
        const int kz = zyx & 255, ky = (zyx >> 8) & 255, kx = zyx >> 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bool ok = tap_ok && iz0[mt] > -(1 << 19);
            if (!no_pad)
                ok = ok && (unsigned)(iz0[mt] + kz) < (unsigned)a.Di && (unsigned)(iy0[mt] + ky) < (unsigned)a.Hi &&
                     (unsigned)(ix0[mt] + kx) < (unsigned)a.Wi;
            const unsigned off = ok ? (unsigned)((pin0[mt] + ofs) * CIN + c) * 4u : 0xFFFFFFF0u;
            A[mt] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            Bv[nt] = *reinterpret_cast<const f32x4v*>(wp + (long)s * wstep + (long)nt * 256);
    };

    auto mma_step = [&](const f32x4v (&A)[MT], const f32x4v (&Bv)[NT]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bv[nt][j], A[mt][j], acc[mt][nt], 0, 0, 0);
    };

    // Two register sets, ping-pong: the loads of step s+1 are in flight under the MFMAs of step s.
    // Every prefetch is unconditional (the last one re-reads the final step) so that hipcc can
    // count the outstanding loads exactly (a conditional prefetch makes it wait vmcnt(0)).
    f32x4v ag[MT], bg[NT];
    if (nsteps > 0) {
        load_step(0, af, bf);
        int s = 0;
        for (; s + 2 <= nsteps; s += 2) {
            load_step(s + 1, ag, bg);
            mma_step(af, bf);
            load_step(s + 2 < nsteps ? s + 2 : nsteps - 1, af, bf);
            mma_step(ag, bg);
        }
        if (s < nsteps) mma_step(af, bf);
    }

    if (SPLITK) {
        if (wave > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) red[((wave - 1) * MT * NT + mt * NT + nt) * 64 + lane] = acc[mt][nt];
        }
        __syncthreads();
        if (wave > 0) return;    // SPLITK grids are exact (one tile group per workgroup): no second iteration
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4v r = red[(w * MT * NT + mt * NT + nt) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mt][nt][j] += r[j];
                }
    }

    // epilogue.  The MFMA operands are swapped (weights in the A slot, activations in the B slot), so
    // the accumulator is D^T: this lane holds 4 CONSECUTIVE output channels (4*lq .. 4*lq+3 of each
    // N tile) of ONE voxel (its A-role voxel, lm) -> one float4 store per tile, 16 lanes x 16 B contiguous.
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (opix[mt] < 0) continue;
        const int b = a.skip_mode == 2 ? opix[mt] / (a.DoF * a.HoF * a.WoF) : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n0 = (nt0 + nt) * 16 + lq * 4;
            if (n0 >= a.cout) continue;
            epilogue_store(a, acc[mt][nt], n0, opix[mt], b, oyx[mt] >> 16, oyx[mt] & 0xffff);
        }
    }
    }   // grid-stride loop over tile groups
}

template <int CIN, int MT, int NT, bool SPLITK>
int launch(const ConvArgs& a, hipStream_t s) {
    const long Mtot = (long)a.B * a.Do * a.Ho * a.Wo;
    if (Mtot >= (1L << 31) || (long)a.B * a.Di * a.Hi * a.Wi >= (1L << 31)) return MVSTER_ERR_SHAPE;
    const long tiles = (Mtot + 15) / 16;
    const long per_block = SPLITK ? MT : 4 * MT;
    long gx = (tiles + per_block - 1) / per_block;
    // (a capped, persistent grid was measured slower on every layer: 436 vs 452 depth-maps/s; the
    //  grid-stride loop in the kernel therefore runs exactly once)
    dim3 grid((unsigned)gx, a.ntile_total / NT, a.nclass);
    MV_NOTE_KERNEL("conv_mfma_kernel<%d, %d, %d, %s>", CIN, MT, NT, SPLITK ? "true" : "false");
    hipLaunchKernelGGL((conv_mfma_kernel<CIN, MT, NT, SPLITK>), grid, dim3(256), 0, s, a);
    return mv_check_launch();
}

template <int CIN>
int dispatch_tiles(const ConvArgs& a, int MT, int NT, bool splitk, hipStream_t s) {
    if (splitk) {
        if constexpr (CIN >= 16) {
#define MV_S(M_, N_) if (MT == M_ && NT == N_) return launch<CIN, M_, N_, true>(a, s);
            MV_S(1, 1) MV_S(1, 2) MV_S(1, 4) MV_S(2, 1) MV_S(2, 2)
#undef MV_S
        }
        return MVSTER_ERR_UNSUPPORTED;
    }
#define MV_T(M_, N_) if (MT == M_ && NT == N_) return launch<CIN, M_, N_, false>(a, s);
    MV_T(1, 1) MV_T(2, 1) MV_T(4, 1) MV_T(1, 2) MV_T(2, 2) MV_T(4, 2) MV_T(1, 4) MV_T(2, 4) MV_T(4, 4)
    if constexpr (CIN == 64) {   // 72 = 9 x 8 and 144 = 9 x 16 output channels of the re-associated FPN levels
        MV_T(1, 5) MV_T(2, 5) MV_T(4, 5)
        MV_T(1, 3) MV_T(2, 3) MV_T(4, 3) MV_T(1, 9) MV_T(2, 9)
    }
#undef MV_T
    return MVSTER_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// LDS-staged variant for ordinary (non-transposed) convolutions with CIN % 16 == 0.
//
// In the direct kernel above every A operand is a global load; a 3x3(x3) layer re-reads each
// input texel 9 (27) times and the per-CU L1 (64 B/clk, one tag lookup per touched line) becomes
// the limiter at ~35 % of the MFMA peak.  Here a workgroup (4 waves) owns a TY x 32 output tile of
// one (b, z) slice, TY = 2*MT.  Per 16-channel chunk of the input it stages the
// (KD) x (TY-1)*s+KH x 31*s+KW input patch once into LDS (zero padding materialised there, so
// the K loop has no bounds logic at all), then every tap is MT ds_read_b128 + NT coalesced weight
// loads feeding 4*MT*NT MFMAs.  Same packed weights, same K order (tap-major, channel-minor) and the
// same fused epilogue as the direct kernel.  Several workgroups per CU overlap staging and math.
// ------------------------------------------------------------------------------------------
// LDS layout of the staged patch: two planes of [pixel][2 quads] (quad = 4 consecutive channels of the 16-channel
// chunk), plane = quad >> 1.  A ds_read_b128 is served in groups of 16 lanes that must cover 16 distinct 16-byte slots
// modulo 256 B; with this kernel's lane = (quad, m) mapping each group holds eight lanes of quad 2k (pixels m) and eight
// of quad 2k+1, i.e. slots 2*(P + m) and 2*(P + m) + 1 of one plane for eight m that are distinct modulo 8: conflict-free
// for stride-1 layers at any tap offset (the previous [pixel][4 quads] layout put m and m + 4 on the same banks: 30-40 %
// of the LDS cycles were conflicts, r01_o PMC).  patch_plane() pads a plane to 4 mod 8 slots, which keeps the two planes
// 16 banks apart for the staging stores (a thread quartet writes quads 0..3 of one pixel).
// Multiply-shift forms of the prologue's divisions (tile decode, patch pixel -> row / column): with plain `/` and `%` the
// prologue spent 11 reciprocal-based division sequences (~220 of its ~570 instructions) per wavefront -- on a one-chunk
// layer the MFMA phase is only 72 instructions long, so the prologue and epilogue are what the MFMA pipes wait for.
struct LdsDivs {
    unsigned mul[6], shr[6];     // tiles_x, tiles_y, Do, PW, PH, KH
};
constexpr int kMaxStage = 12;   // float4 loads per thread per chunk (patch <= 48 KB)
__device__ __forceinline__ int patch_plane(int npix) { return ((npix * 2 + 7) & ~7) + 4; }   // float4 units, = 4 mod 8
static const bool g_no_wlds = MV_PROBE_ENV("MVSTER_NO_WLDS") != nullptr;   // experiment switch: weights from L1 again

// WN > 0: the chunk's weights are staged in LDS too, WN float4 per thread (taps * NT * 64 <= WN * 256).
// Used with WN = 3 (2-D 3x3, NT = 1); WN = 7 (3x3x3, 28 KB) was measured slower: it costs a workgroup of occupancy.
template <int MT, int NT, int KW, int NG, int WN>
__global__ void __launch_bounds__(256) conv_lds_kernel(ConvArgs a, int tiles_x, int tiles_y, LdsDivs dv) {
    extern __shared__ __attribute__((aligned(16))) float patch_raw[];
    MV_TL(0);
    MV_TL_ID();
    f32x4v* patch_base = reinterpret_cast<f32x4v*>(patch_raw);
    f32x4v* wl = patch_base + NG * 1024 + 32;     // WL: this chunk's packed weights [tap][nt][lane] (after the plane pads)
    constexpr int TY = 2 * MT;
    const int KD = a.kd[0], KH = a.kh[0];
    const int PW = 31 * a.sw + KW, PH = (TY - 1) * a.sh + KH;
    const int CIN = a.cin, nchunks = CIN >> 4;
    const int plane = patch_plane(KD * PH * PW);

    unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    unsigned q = fast_div(bid, tiles_x, dv.mul[0], dv.shr[0]);
    const int tile_x = bid - q * tiles_x; bid = q;
    q = fast_div(bid, tiles_y, dv.mul[1], dv.shr[1]);
    const int tile_y = bid - q * tiles_y; bid = q;
    q = fast_div(bid, a.Do, dv.mul[2], dv.shr[2]);
    const int zo = bid - q * a.Do;
    const int b = q;
    const int ty0 = tile_y * TY, tx0 = tile_x * 32;
    const int nt0 = blockIdx.y * NT;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lm = lane & 15, lq = lane >> 4;

    // chunk-independent global element offset of each patch pixel this thread stages (-1 = zero padding): a thread owns
    // all four quads of its pixels, so the pixel -> (row, column) arithmetic and the bounds checks are done once per pixel
    // Stride-2 layers keep the even columns of a patch row first, then the odd ones: a lane's A operand for tap kx is
    // column 2*(xs*16 + lm) + kx, so with the plain layout the 16 lanes of a ds_read_b128 are 64 bytes apart (a 2-way bank
    // conflict on every read: SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles of the 32 -> 64 5x5 layer, r03 / r04 PMC); split
    // by parity they are 32 bytes apart, as in a stride-1 layer.
    const bool split = a.sw == 2;
    const int PWH = (PW + 1) >> 1;
    const int npix = KD * PH * PW;
    int goff[NG], gslot[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const unsigned pix = threadIdx.x + g * 256;
        int off = -1;
        gslot[g] = (int)pix;
        if ((int)pix < npix) {
            const unsigned prow = fast_div(pix, PW, dv.mul[3], dv.shr[3]);       // = pz * PH + py
            const int px = pix - prow * PW;
            if (split) gslot[g] = (int)(prow * PW) + (px & 1) * PWH + (px >> 1);
            const int pz = fast_div(prow, PH, dv.mul[4], dv.shr[4]);
            const int py = prow - pz * PH;
            const int iz = zo * a.sd - a.pd[0] + pz, iy = ty0 * a.sh - a.ph[0] + py, ix = tx0 * a.sw - a.pw[0] + px;
            if ((unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi)
                off = (((b * a.Di + iz) * a.Hi + iy) * a.Wi + ix) * CIN;
        }
        goff[g] = off;
    }
    const long zero_off = a.zeros - a.in;

    const float* wp = a.wpk + ((long)nt0 * 64 + lane) * 4;
    const long wstep = (long)a.ntile_total * 256;

    // Staging is register-double-buffered: NG groups of 4 float4 per thread.  The global loads of chunk
    // ch+1 are issued BEFORE the MFMAs of chunk ch (their latency hides under this workgroup's own math
    // instead of relying on other workgroups being out of phase) and written to LDS after them.
    // WL (small tap count x NT): the chunk's weights are staged the same way, so that inside the MFMA loop
    // both operands come from LDS (~100 cycles) instead of one of them from L1/L2 (~500+ under load), which
    // the one-row-deep software pipeline cannot cover.
    f32x4v stg[NG * 4];
    constexpr bool WL = WN > 0;
    f32x4v wst[WL ? WN : 1];
    const int nW = KD * KH * KW * NT * 64;
    auto stage_load = [&](int ch) {
#pragma unroll
        for (int i = 0; i < NG * 4; ++i) {
            const int o = goff[i >> 2];
            const long off = o >= 0 ? (long)o + ch * 16 + (i & 3) * 4 : zero_off;
            stg[i] = *reinterpret_cast<const f32x4v*>(a.in + off);
        }
        if (WL) {
#pragma unroll
            for (int i = 0; i < WN; ++i) {
                const int idx = min((int)threadIdx.x + i * 256, nW - 1);
                const int t = idx / (NT * 64), rem = idx - t * (NT * 64);
                wst[i] = *reinterpret_cast<const f32x4v*>(a.wpk + (long)(t * nchunks + ch) * wstep + ((long)nt0 * 64 + rem) * 4);
            }
        }
    };
    auto stage_store = [&](f32x4v* dst) {
#pragma unroll
        for (int i = 0; i < NG * 4; ++i) {
            const int pix = threadIdx.x + (i >> 2) * 256, quad = i & 3;
            if (pix < npix) dst[(quad >> 1) * plane + gslot[i >> 2] * 2 + (quad & 1)] = stg[i];
        }
        if (WL) {
#pragma unroll
            for (int i = 0; i < WN; ++i) wl[threadIdx.x + i * 256] = wst[i];
        }
    };
    stage_load(0);
    // everything below up to the LDS store is independent of the loads just issued and runs under their latency
    // LDS float4 index of this lane's A operand for tap (0,0,0) of each of its M tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, xs = t & 1;
        abase[mt] = ((row * a.sh) * PW + (xs * 16 + lm) * (split ? 1 : a.sw)) * 2 + (lq >> 1) * plane + (lq & 1);
    }
    int kxo[KW];          // float4 offset of tap column kx within a patch row (wave-uniform)
#pragma unroll
    for (int kx = 0; kx < KW; ++kx) kxo[kx] = split ? ((kx & 1) * PWH + (kx >> 1)) * 2 : kx * 2;

    f32x4v acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    // BatchNorm scale / shift of this lane's output channels, fetched ahead of the MFMAs instead of after them
    f32x4v scv[NT], shv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = min((nt0 + nt) * 16 + lq * 4, a.ntile_total * 16 - 4);
        scv[nt] = *reinterpret_cast<const f32x4v*>(a.scale + n0);
        shv[nt] = *reinterpret_cast<const f32x4v*>(a.shift + n0);
    }
    MV_TL(1);
    stage_store(patch_base);
    MV_TL(2);
    __syncthreads();
    MV_TL(3);
    for (int ch = 0; ch < nchunks; ++ch) {
        const f32x4v* patch = patch_base;
        if (ch + 1 < nchunks) stage_load(ch + 1);
        // One "row" = the KW taps of one (kz, ky).  Rows are software-pipelined with two register sets:
        // the LDS reads and weight loads of row r+1 are issued before the 4*KW*MT*NT MFMAs of row r, and
        // every prefetch is unconditional so that hipcc emits counted waits.
        // depth taps that fall entirely into the zero padding of this output slice are skipped (their staged
        // rows are zeros: 2 of the 12 (slice, kz) pairs of a 3x3x3 layer on a 4-deep volume)
        const int kz_lo = max(0, a.pd[0] - zo * a.sd), kz_hi = min(KD, a.Di + a.pd[0] - zo * a.sd);
        const int r_first = kz_lo * KH, nrows = kz_hi * KH;
        auto load_row = [&](int r, f32x4v (&A)[KW][MT], f32x4v (&Bv)[KW][NT]) {
            const int kz = fast_div(r, KH, dv.mul[5], dv.shr[5]), ky = r - kz * KH;   // (no division sequence per row of taps)
            const int rowoff = (kz * PH + ky) * PW * 2;
            const float* w = wp + (long)(r * KW * nchunks + ch) * wstep;
#pragma unroll
            for (int kx = 0; kx < KW; ++kx) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) A[kx][mt] = patch[abase[mt] + rowoff + kxo[kx]];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    Bv[kx][nt] = WL ? wl[((r * KW + kx) * NT + nt) * 64 + lane]
                                    : *reinterpret_cast<const f32x4v*>(w + (long)kx * nchunks * wstep + nt * 256);
            }
        };
        auto mma_row = [&](const f32x4v (&A)[KW][MT], const f32x4v (&Bv)[KW][NT]) {
#pragma unroll
            for (int kx = 0; kx < KW; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bv[kx][nt][j], A[kx][mt][j], acc[mt][nt], 0, 0, 0);
        };
        f32x4v a0[KW][MT], b0[KW][NT], a1[KW][MT], b1[KW][NT];
        load_row(r_first, a0, b0);
        int r = r_first;
        for (; r + 2 <= nrows; r += 2) {
            load_row(r + 1, a1, b1);
            mma_row(a0, b0);
            load_row(r + 2 < nrows ? r + 2 : nrows - 1, a0, b0);
            mma_row(a1, b1);
        }
        if (r < nrows) mma_row(a0, b0);
        if (ch + 1 < nchunks) {
            __syncthreads();                  // everyone is done reading chunk ch
            stage_store(patch_base);
            __syncthreads();
        }
    }

    MV_TL(4);
    // epilogue (swapped operands: this lane holds channels 4*lq..4*lq+3 of its own voxel, see above)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int y = ty0 + (t >> 1), x = tx0 + (t & 1) * 16 + lm;
        if (y >= a.Ho || x >= a.Wo) continue;
        const long opix = (((long)b * a.Do + zo) * a.Ho + y) * a.Wo + x;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n0 = (nt0 + nt) * 16 + lq * 4;
            if (n0 >= a.cout) continue;
            epilogue_store(a, acc[mt][nt], n0, opix, b, y, x, scv[nt], shv[nt]);
        }
    }
    MV_TL(5);
}

template <int MT, int NT, int KW, int NG>
int launch_lds_ng(const ConvArgs& a, int tiles_x, int tiles_y, hipStream_t s) {
    const long blocks = (long)tiles_x * tiles_y * a.Do * a.B;
    if (blocks >= (1L << 31) || (long)a.B * a.Di * a.Hi * a.Wi * a.cin >= (1L << 31)) return MVSTER_ERR_SHAPE;
    dim3 grid((unsigned)blocks, a.ntile_total / NT, 1);
    LdsDivs dv;
    const unsigned divisors[6] = {(unsigned)tiles_x, (unsigned)tiles_y, (unsigned)a.Do, (unsigned)(31 * a.sw + KW),
                                  (unsigned)((2 * MT - 1) * a.sh + a.kh[0]), (unsigned)a.kh[0]};
    for (int i = 0; i < 6; ++i) find_divisor(divisors[i], dv.mul[i], dv.shr[i]);
    const int nw = a.kd[0] * a.kh[0] * KW * NT * 64;       // weight float4 per chunk
    if (nw <= 3 * 256 && !g_no_wlds) {
        const size_t lds = (size_t)(NG * 1024 + 32) * 16 + 3 * 256 * 16;
        MV_NOTE_KERNEL("conv_lds_kernel<%d, %d, %d, %d, 3>", MT, NT, KW, NG);
        hipLaunchKernelGGL((conv_lds_kernel<MT, NT, KW, NG, 3>), grid, dim3(256), lds, s, a, tiles_x, tiles_y, dv);
    } else {
        const size_t lds = (size_t)(NG * 1024 + 32) * 16;
        MV_NOTE_KERNEL("conv_lds_kernel<%d, %d, %d, %d, 0>", MT, NT, KW, NG);
        hipLaunchKernelGGL((conv_lds_kernel<MT, NT, KW, NG, 0>), grid, dim3(256), lds, s, a, tiles_x, tiles_y, dv);
    }
    return mv_check_launch();
}

template <int MT, int NT, int KW>
int launch_lds(const ConvArgs& a, hipStream_t s) {
    constexpr int TY = 2 * MT;
    const int PW = 31 * a.sw + KW, PH = (TY - 1) * a.sh + a.kh[0];
    const size_t nstage = (size_t)a.kd[0] * PH * PW * 4;
    if (nstage > (size_t)kMaxStage * 256) return MVSTER_ERR_SHAPE;
    const int tiles_x = (a.Wo + 31) / 32, tiles_y = (a.Ho + TY - 1) / TY;
    const int ng = (int)((nstage + 1023) / 1024);
    if (ng == 1) return launch_lds_ng<MT, NT, KW, 1>(a, tiles_x, tiles_y, s);
    if (ng == 2) return launch_lds_ng<MT, NT, KW, 2>(a, tiles_x, tiles_y, s);
    return launch_lds_ng<MT, NT, KW, 3>(a, tiles_x, tiles_y, s);
}

int dispatch_lds(const ConvArgs& a, int MT, int NT, hipStream_t s) {
    if (a.nclass != 1 || a.cin % 16 != 0 || a.skip_mode > 1 || a.osd != 1 || a.osh != 1 || a.osw != 1)
        return MVSTER_ERR_UNSUPPORTED;
#define MV_L(M_, N_, K_) if (MT == M_ && NT == N_ && a.kw[0] == K_) return launch_lds<M_, N_, K_>(a, s);
    MV_L(2, 1, 3) MV_L(2, 2, 3) MV_L(2, 4, 3) MV_L(4, 1, 3) MV_L(4, 2, 3) MV_L(4, 4, 3)
    MV_L(2, 1, 5) MV_L(2, 2, 5) MV_L(2, 4, 5) MV_L(4, 1, 5) MV_L(4, 2, 5) MV_L(4, 4, 5)
#undef MV_L
    return MVSTER_ERR_UNSUPPORTED;
}

// one raw MFMA, to pin the fragment layout this file assumes (tests/test_gpu_conv.py)
__global__ void mfma_probe_kernel(const float* A, const float* Bm, float* Dm) {
    const int lane = threadIdx.x;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    const float av = A[(lane & 15) * 4 + (lane >> 4)];    // A[m][k], 16x4 row major
    const float bv = Bm[(lane >> 4) * 16 + (lane & 15)];  // B[k][n], 4x16 row major
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) Dm[((lane >> 4) * 4 + r) * 16 + (lane & 15)] = acc[r];
}

}  // namespace

// geom: int32 array, see mvster_amd/conv_plan.py (GEOM_* layout); woff: per-class offsets (floats)
extern "C" int mvster_conv_mfma(const float* in, const float* wpk, const float* scale, const float* shift,
                                const float* skip, const float* zeros, const float* prob_w, const float* prob_b,
                                float* out, const int* geom, int ngeom, const long* woff, int cin, int mt, int nt,
                                int variant, void* stream) {
    if (!in || !wpk || !scale || !shift || !zeros || !out || !geom || !woff) return MVSTER_ERR_NULL;
    if (ngeom < 22) return MVSTER_ERR_SHAPE;
    ConvArgs a;
    a.in = in; a.wpk = wpk; a.scale = scale; a.shift = shift; a.skip = skip; a.zeros = zeros; a.out = out;
    a.prob_w = prob_w; a.prob_b = prob_b;
    if ((prob_w == nullptr) != (prob_b == nullptr)) return MVSTER_ERR_NULL;
    int i = 0;
    a.B = geom[i++]; a.Di = geom[i++]; a.Hi = geom[i++]; a.Wi = geom[i++];
    a.Do = geom[i++]; a.Ho = geom[i++]; a.Wo = geom[i++];
    a.DoF = geom[i++]; a.HoF = geom[i++]; a.WoF = geom[i++];
    a.sd = geom[i++]; a.sh = geom[i++]; a.sw = geom[i++];
    a.cout = geom[i++]; a.ntile_total = geom[i++]; a.relu = geom[i++]; a.skip_mode = geom[i++];
    a.osd = geom[i++]; a.osh = geom[i++]; a.osw = geom[i++];
    a.nclass = geom[i++];
    if (a.nclass < 1 || a.nclass > kMaxClasses || ngeom < i + a.nclass * 10) return MVSTER_ERR_SHAPE;
    for (int c = 0; c < a.nclass; ++c) {
        a.kd[c] = geom[i++]; a.kh[c] = geom[i++]; a.kw[c] = geom[i++];
        a.pd[c] = geom[i++]; a.ph[c] = geom[i++]; a.pw[c] = geom[i++];
        a.od[c] = geom[i++]; a.oh[c] = geom[i++]; a.ow[c] = geom[i++];
        a.nsteps[c] = geom[i++];
        a.woff[c] = woff[c];
        if (a.kd[c] * a.kh[c] * a.kw[c] > kMaxTaps || a.nsteps[c] < 1) return MVSTER_ERR_SHAPE;
    }
    if (a.B <= 0 || a.Do <= 0 || a.Ho <= 0 || a.Wo <= 0 || a.cout <= 0 || a.ntile_total <= 0) return MVSTER_ERR_SHAPE;
    find_divisor((unsigned)a.Wo, a.div_mul[0], a.div_shr[0]);
    find_divisor((unsigned)a.Ho, a.div_mul[1], a.div_shr[1]);
    find_divisor((unsigned)a.Do, a.div_mul[2], a.div_shr[2]);
    {
        const long in_elems = (long)a.B * a.Di * a.Hi * a.Wi * cin;
        const long out_elems = (long)a.B * a.DoF * a.HoF * a.WoF * a.cout;
        if (in_elems >= (1L << 30) || out_elems >= (1L << 31)) return MVSTER_ERR_SHAPE;   // 32-bit offsets
        a.in_bytes = (unsigned)(in_elems * 4);
    }
    for (int c = 0; c < a.nclass; ++c) {
        // i = o*s - p + k stays inside [0, extent) for every lattice point and tap?
        auto inside = [](int n_out, int s, int p, int k, int extent) {
            return -p >= 0 && (n_out - 1) * s - p + (k - 1) < extent;
        };
        a.all_inside[c] = inside(a.Do, a.sd, a.pd[c], a.kd[c], a.Di) && inside(a.Ho, a.sh, a.ph[c], a.kh[c], a.Hi) &&
                          inside(a.Wo, a.sw, a.pw[c], a.kw[c], a.Wi);
    }
    if (a.skip_mode != 0 && !skip) return MVSTER_ERR_NULL;
    if (a.ntile_total % nt != 0) return MVSTER_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    a.cin = cin;
    if ((variant & 0xff) == 5) return dispatch_pers(a, mt, nt, variant >> 8, s);   // persistent LDS-DMA family (conv_pers.hip)
    if ((variant & 0xff) == 6) return dispatch_1x1(a, mt, variant >> 8, s);        // persistent 1x1, weights in LDS
    if ((variant & 0xff) == 7) return dispatch_pp(a, mt, nt, s);                   // persistent, eight waves in ping-pong
    if ((variant & 0xff) == 8) return dispatch_wino(a, nt, variant >> 8, false, s);   // Winograd F(2x2,3x3); wpk = transformed weights
    if ((variant & 0xff) == 9) return dispatch_wino(a, nt, variant >> 8, true, s);   // ... deep layers: slices and weights streamed
    if ((variant & 0xff) == 11) return dispatch_b3(a, mt, variant >> 8, s);                       // 3 x bf16-split operands on the bf16 MFMA (conv_b3.hip)
    if (prob_w && (a.cout != 8 || a.skip_mode == 2 || variant == 1)) return MVSTER_ERR_UNSUPPORTED;
    if (variant == 1) return dispatch_lds(a, mt, nt, s);
    if (variant != 0 && variant != 2) return MVSTER_ERR_UNSUPPORTED;
    const bool splitk = variant == 2;
    switch (cin) {
        case 4: return dispatch_tiles<4>(a, mt, nt, splitk, s);
        case 8: return dispatch_tiles<8>(a, mt, nt, splitk, s);
        case 16: return dispatch_tiles<16>(a, mt, nt, splitk, s);
        case 32: return dispatch_tiles<32>(a, mt, nt, splitk, s);
        case 64: return dispatch_tiles<64>(a, mt, nt, splitk, s);
        case 80: return dispatch_tiles<80>(a, mt, nt, splitk, s);   // 72 gradient channels of the FPN gather, zero-padded
        default: return MVSTER_ERR_UNSUPPORTED;
    }
}

#ifdef MVSTER_TIMELINE
extern "C" int mvster_debug_timeline(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl), &buf, sizeof(buf)) == hipSuccess ? MVSTER_OK : MVSTER_ERR_LAUNCH;
}
#endif

extern "C" int mvster_mfma_probe(const float* A, const float* B, float* D, void* stream) {
    if (!A || !B || !D) return MVSTER_ERR_NULL;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
    return mv_check_launch();
}

// Packed-weight refresh on the device (training: once per layer and optimizer step).  Writes the MFMA fragment
// order [K/16][N/16][64 lanes][4] (mvster_amd/conv_plan.py:_pack_gemm) of the implicit-GEMM B matrix
//   Bm[k = tap*cin_pad + ci][n] = w[n*s_n + ci*s_c + kz*s_z + ky*s_y + kx*s_x]     (taps optionally flipped)
// straight from the parameter tensor: one launch instead of the permute / pad / reshape / cat chain.
namespace {
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wpk, int nsteps,
                                                           int ntile, int cout, int cin, int cin_pad, int kd, int kh, int kw,
                                                           long s_n, long s_c, long s_z, long s_y, long s_x, int flip) {
    const int idx = blockIdx.x * 256 + threadIdx.x;          // one float4 of the packed array
    if (idx >= nsteps * ntile * 64) return;
    const int lane = idx & 63, t = (idx >> 6) % ntile, st = (idx >> 6) / ntile;
    const int n = t * 16 + (lane & 15);
    const int ntaps = kd * kh * kw;
    f32x4v out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = st * 16 + (lane >> 4) * 4 + j;
        const int tap = k / cin_pad, ci = k - tap * cin_pad;
        float v = 0.0f;
        if (tap < ntaps && ci < cin && n < cout) {
            int kx = tap % kw, ky = (tap / kw) % kh, kz = tap / (kw * kh);
            if (flip) { kx = kw - 1 - kx; ky = kh - 1 - ky; kz = kd - 1 - kz; }
            v = w[(long)n * s_n + (long)ci * s_c + (long)kz * s_z + (long)ky * s_y + (long)kx * s_x];
        }
        out[j] = v;
    }
    *reinterpret_cast<f32x4v*>(wpk + (long)idx * 4) = out;
}
}  // namespace

namespace {
// Transposed layers: one workgroup row (blockIdx.y) per output-parity class; the class's kernel taps are a short list
// of flattened (kz, ky, kx) indices into w [cin, cout, kd*kh*kw] (contiguous), its packed block starts at woff.
struct PackClasses {
    int nclass;
    int ntaps[kMaxClasses];
    long woff[kMaxClasses];
    unsigned char tap[kMaxClasses][27];
};

__global__ void __launch_bounds__(256) pack_weights_classes_kernel(const float* __restrict__ w, float* __restrict__ wpk,
                                                                   PackClasses pc, int ntile, int cout, int cin, int cin_pad,
                                                                   int ktot) {
    const int cls = blockIdx.y;
    const int ntaps = pc.ntaps[cls];
    const int nsteps = (ntaps * cin_pad + 15) / 16;
    const int idx = blockIdx.x * 256 + threadIdx.x;          // one float4 of the class's packed block
    if (idx >= nsteps * ntile * 64) return;
    const int lane = idx & 63, t = (idx >> 6) % ntile, st = (idx >> 6) / ntile;
    const int n = t * 16 + (lane & 15);
    f32x4v out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = st * 16 + (lane >> 4) * 4 + j;
        const int tl = k / cin_pad, ci = k - tl * cin_pad;
        float v = 0.0f;
        if (tl < ntaps && ci < cin && n < cout) v = w[((long)ci * cout + n) * ktot + pc.tap[cls][tl]];
        out[j] = v;
    }
    *reinterpret_cast<f32x4v*>(wpk + pc.woff[cls] + (long)idx * 4) = out;
}
}  // namespace

// w [cin, cout, kd, kh, kw] contiguous (ConvTranspose layout, or a conv weight [cout_fwd, cin_fwd, ...] used as the
// input-gradient operator); taps [nclass][27] (flattened tap indices, class c uses the first ntaps[c]); woff [nclass]
// float offsets of the classes' blocks in wpk (mvster_amd/conv_plan.py:_class_table).
extern "C" int mvster_pack_conv_weights_classes(const float* w, float* wpk, int cout, int cin, int cin_pad, int ktot, int nclass,
                                                const int* ntaps, const long* woff, const int* taps, void* stream) {
    if (!w || !wpk || !ntaps || !woff || !taps) return MVSTER_ERR_NULL;
    if (cout <= 0 || cin <= 0 || cin_pad < cin || ktot <= 0 || ktot > 27 || nclass <= 0 || nclass > kMaxClasses)
        return MVSTER_ERR_SHAPE;
    PackClasses pc;
    pc.nclass = nclass;
    int maxsteps = 0;
    for (int c = 0; c < nclass; ++c) {
        if (ntaps[c] <= 0 || ntaps[c] > 27) return MVSTER_ERR_SHAPE;
        pc.ntaps[c] = ntaps[c];
        pc.woff[c] = woff[c];
        for (int t = 0; t < 27; ++t) pc.tap[c][t] = (unsigned char)(t < ntaps[c] ? taps[c * 27 + t] : 0);
        const int ns = (ntaps[c] * cin_pad + 15) / 16;
        maxsteps = ns > maxsteps ? ns : maxsteps;
    }
    const int ntile = (cout + 15) / 16;
    const int total = maxsteps * ntile * 64;
    hipLaunchKernelGGL(pack_weights_classes_kernel, dim3((total + 255) / 256, nclass), dim3(256), 0, (hipStream_t)stream, w, wpk,
                       pc, ntile, cout, cin, cin_pad, ktot);
    return mv_check_launch();
}

extern "C" int mvster_pack_conv_weights(const float* w, float* wpk, int cout, int cin, int cin_pad, int kd, int kh, int kw,
                                        long s_n, long s_c, long s_z, long s_y, long s_x, int flip, void* stream) {
    if (!w || !wpk) return MVSTER_ERR_NULL;
    if (cout <= 0 || cin <= 0 || cin_pad < cin || kd <= 0 || kh <= 0 || kw <= 0) return MVSTER_ERR_SHAPE;
    const int nsteps = (kd * kh * kw * cin_pad + 15) / 16, ntile = (cout + 15) / 16;
    const int total = nsteps * ntile * 64;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, wpk, nsteps, ntile,
                       cout, cin, cin_pad, kd, kh, kw, s_n, s_c, s_z, s_y, s_x, flip);
    return mv_check_launch();
}
