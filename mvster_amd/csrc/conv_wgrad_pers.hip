// Weight gradient of the (1|3)x3x3 stride-1 convolutions, persistent LDS-DMA form.
//
//   dW[tap][co][ci] = sum over output voxels o of  gy[o][co] * x[o - p + tap][ci]        (conv_wgrad.hip has the general case)
//
// conv_wgrad_lds_kernel stages one 64-column chunk of one output row at a time through registers and two barriers; a
// wavefront then has 36 MFMAs (16 -> 16 channels, 9 taps) before it waits again, and the kernel depends on co-resident
// workgroups to hide that: 0.23 of the fp32 MFMA peak over the weight gradients of a training step
// (profiles/r03_l_train_bench.json).  Here a workgroup walks its share of UNITS of R output rows x 64 columns:
//   * the unit's gy block and the (R + 2) x kd input rows it meets arrive by LDS-DMA (buffer_load ... lds: contiguous
//     16-pixel x 16-channel kilobytes, zero padding = out-of-range offsets) into the second LDS buffer while the MFMAs of the
//     current unit run; one barrier per unit;
//   * operands are read as in the staged kernel: the voxels are the K dimension, lane (r = lane & 15, k = lane >> 4) of
//     a K step of four consecutive columns reads gy[px + k][co0 + r] and x[px + k + tap][ci0 + r] with ds_read_b32 from
//     dense [pixel][16 channels] planes (the four pixels of a K step are 64 bytes apart: one 256-byte row of banks);
//   * MT x NT channel tiles x all taps of accumulators live in registers across the units; four compute waves split a
//     unit's K steps and meet in LDS once, at the end; same `partial` slots as the other kernels (the host-side finish adds
//     them in a fixed order: deterministic);
//   * waves 4-7 only issue the LDS-DMA: a wave that asks for data at the rate HBM delivers it (25 B/ns per CU,
//     scripts/probes/lds_dma_bw.hip) stalls ~300 cycles per kilobyte at issue -- measured with the requests in the compute
//     waves: 19 us of streaming and 40 us of MFMAs ADD UP (16 -> 16 at 10 x 256 x 320); in waves of their own they overlap.
// Autograd of nn.Conv2d / nn.Conv3d (models/mvs4net_utils.py:116-123, :224-251) as used by FPN4 and reg2d.
#include <stdlib.h>

#include "conv_args.hpp"
#include "conv_wgrad.hpp"

namespace mvwgrad {
// (timing experiments of the probe build: bit 0 drops the MFMAs, bit 1 the loads; the product kernel has neither switch)
#ifdef MVSTER_PROBES
#define MV_WG_DBG(bit) (dbg & (bit))
#else
#define MV_WG_DBG(bit) false
#endif

namespace {

using mvconv::f32x4v;
using mvconv::lds_void;

// XC = output columns per unit (a multiple of 16): 64, or 80 / 48 / 32 where the map's width leaves a 64-column grid mostly
// empty (W = 80: two chunks of 64 are 38 % zero columns -- MFMAs on zeros; one chunk of 80 has none).  The staged input rows
// are XC + 16 columns (XC + 2 needed, in 16-pixel pieces).
template <int MT, int NT, int KD, int R, int XC>
__global__ void __launch_bounds__(512) conv_wgrad_pers_kernel(WgradArgs a, int mgroups, int nblk, unsigned x_bytes, unsigned gy_bytes, int dbg) {
    constexpr int TAPS = KD * 9;
    constexpr int kXC = XC, kXBP = XC + 16;
    constexpr int PA = XC / 16, PB = kXBP / 16;             // 16-pixel pieces per staged gy / input row
    static_assert(XC % 16 == 0 && (R * (XC / 4)) % 4 == 0, "whole pieces, and the four compute waves split the K steps evenly");
    constexpr int RB = R + 2;                               // input rows per depth slice
    constexpr int APL = R * kXC * 4, BPL = KD * RB * kXBP * 4;            // float4 per plane
    constexpr int BUF = MT * APL + NT * BPL;                // float4 per buffer
    constexpr int NPA = MT * R * PA, NPB = NT * KD * RB * PB;
    constexpr int NPW = (NPA + NPB + 3) / 4;                // DMA pieces per wave and unit
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const scratch = lds + 2 * BUF;                  // 64 float4: surplus pieces

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;                             // index among the compute waves / among the loading waves
    const bool loader = wave8 >= 4;
    const int active = gridDim.x;                           // workgroups that walk units (<= nblk slots)
    const int r16 = lane & 15, k = lane >> 4;
    const int m0 = (blockIdx.y % mgroups) * MT * 16, n0 = (blockIdx.y / mgroups) * NT * 16;
    const int cop = ((a.CO + 15) / 16) * 16, cipw = ((a.CI + 15) / 16) * 16;
    float* const slot = a.partial + (long)blockIdx.x * TAPS * cop * cipw;        // [TAPS][cop][cipw]

    f32x4v acc[MT][NT][TAPS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[i][j][t] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    {
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), (short)0, (int)x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gy), (short)0, (int)gy_bytes, 0x00020000);
        // units: (b, zo, block of R rows, chunk of 64 columns), chunk fastest
        const int nchunks = (a.Wo + kXC - 1) / kXC, nyb = (a.Ho + R - 1) / R;
        const int nunits = a.B * a.Do * nyb * nchunks;
        const int lp = lane >> 2, lq = lane & 3;            // pixel and channel quad of this lane inside a piece
        auto request = [&](int u, int buf) {
            const int ch = u % nchunks;
            int t = u / nchunks;
            const int yb = t % nyb;
            t /= nyb;
            const int zo = t % a.Do, b = t / a.Do;
            const int x1 = ch * kXC, yo = yb * R;
            f32x4v* const dst0 = lds + buf * BUF;
#pragma unroll 1
            for (int n = 0; n < NPW; ++n) {                 // (rolled: the decode below is scalar work, once per unit)
                const int i = wave + 4 * n;                 // wave-uniform piece index
                unsigned off = 0x80000000u;
                f32x4v* dst = scratch;
                bool is_a = false;
                if (i < NPA) {
                    const int plane = i / (R * PA), rem = i - plane * (R * PA), rr = rem / PA, pb = rem - rr * PA;
                    const int xo = x1 + pb * 16 + lp, y = yo + rr;
                    const bool ok = xo < a.Wo && y < a.Ho;
                    const unsigned o = (unsigned)(((((b * a.Do + zo) * a.Ho + y) * a.Wo + xo) * a.CO + m0 + plane * 16 + lq * 4) * 4);
                    off = ok ? o : 0x80000000u;
                    dst = dst0 + plane * APL + (rr * kXC + pb * 16) * 4;
                    is_a = true;
                } else if (i < NPA + NPB) {
                    const int j = i - NPA;
                    const int plane = j / (KD * RB * PB), rem = j - plane * (KD * RB * PB), row = rem / PB, pb = rem - row * PB;
                    const int kz = row / RB, ry = row - kz * RB;
                    const int p = pb * 16 + lp;
                    const int ix = x1 - a.pw + p, iy = yo - a.ph + ry, iz = zo - a.pd + kz;
                    const bool ok = p < kXC + 2 && (unsigned)ix < (unsigned)a.Wi && (unsigned)iy < (unsigned)a.Hi && (unsigned)iz < (unsigned)a.Di;
                    const unsigned o = (unsigned)(((((b * a.Di + iz) * a.Hi + iy) * a.Wi + ix) * a.CI + n0 + plane * 16 + lq * 4) * 4);
                    off = ok ? o : 0x80000000u;
                    dst = dst0 + MT * APL + plane * BPL + (row * kXBP + pb * 16) * 4;
                }
                // (named operands: hipcc 7.2 drops the kernel's host stub when this builtin gets an expression as its offset)
                if (is_a) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(g_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                }
            }
        };
        auto compute = [&](int buf) {
            const float* const As = reinterpret_cast<const float*>(lds + buf * BUF);
            const float* const Bs = As + MT * APL * 4;
            // K steps of this wave: four consecutive columns of one of the unit's rows each (ks = wave + 4 s).  The operands
            // of step s + 1 are read while the MFMAs of step s run: with one wavefront per SIMD nothing else hides the LDS
            // latency (rolled, a step was 10 reads, a wait, 9 MFMAs: 456 cycles for 288 of MFMA work).
            constexpr int NS = R * (kXC / 4) / 4;                          // steps per wave
            float av[2][MT], bv[2][NT][TAPS];
            auto read_step = [&](int sidx, float (&aa)[MT], float (&bb)[NT][TAPS]) {
                const int ks = wave + 4 * sidx;
                const int rr = ks / (kXC / 4), px = (ks - rr * (kXC / 4)) * 4 + k;
#pragma unroll
                for (int i = 0; i < MT; ++i) aa[i] = As[i * APL * 4 + (rr * kXC + px) * 16 + r16];
                const float* const bp = Bs + (rr * kXBP + px) * 16 + r16;
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int kz = 0; kz < KD; ++kz)
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx)
                                bb[j][(kz * 3 + ky) * 3 + kx] = bp[j * BPL * 4 + ((kz * RB + ky) * kXBP + kx) * 16];
            };
            read_step(0, av[0], bv[0]);
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) {
                if (sidx + 1 < NS) read_step(sidx + 1, av[(sidx + 1) & 1], bv[(sidx + 1) & 1]);
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
                            acc[i][j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sidx & 1][i], bv[sidx & 1][j][t], acc[i][j][t], 0, 0, 0);
            }
        };
        int u = blockIdx.x;
        if (loader && u < nunits) request(u, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; u < nunits; u += active, ++it) {
            const int cur = it & 1;
            if (loader) {
                if (u + active < nunits && !MV_WG_DBG(2)) request(u + active, cur ^ 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the next unit has landed
            } else {
                if (!MV_WG_DBG(1)) compute(cur);
                __builtin_amdgcn_s_waitcnt(0xc07f);                              // done reading this unit
            }
            __builtin_amdgcn_s_barrier();
        }
    }

    // cross-wave sum through LDS (waves 1-3 park their tiles, wave 0 adds in a fixed order), then this workgroup's slot and
    // zeros into the slots no workgroup walks for (every slot of `partial` is written; the finish kernel adds them all)
    constexpr int TILES = MT * NT * TAPS;
    f32x4v* const red = lds;                                 // [3][TILES][64]: inside the two unit buffers
    static_assert(3 * TILES * 64 <= 2 * BUF, "the reduction fits the unit buffers");
    __syncthreads();
    if (!loader && wave > 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int t = 0; t < TAPS; ++t) red[((wave - 1) * TILES + (i * NT + j) * TAPS + t) * 64 + lane] = acc[i][j][t];
    }
    __syncthreads();
    if (wave8 == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    f32x4v sum = acc[i][j][t];
#pragma unroll
                    for (int w = 0; w < 3; ++w) sum += red[(w * TILES + (i * NT + j) * TAPS + t) * 64 + lane];
                    float* out = slot + (long)t * cop * cipw;
#pragma unroll
                    for (int q = 0; q < 4; ++q) out[(m0 + i * 16 + 4 * k + q) * cipw + n0 + j * 16 + r16] = sum[q];
                }
    } else {
        // waves 1-7: this workgroup's (MT*16) x (NT*16) block of every tap in the unwalked slots
        constexpr int PER = TAPS * MT * 16 * NT * 16;
        for (int sl = blockIdx.x + active; sl < nblk; sl += active) {
            float* const z = a.partial + (long)sl * TAPS * cop * cipw;
            for (int e = threadIdx.x - 64; e < PER; e += 448) {
                const int col = e % (NT * 16), row = (e / (NT * 16)) % (MT * 16), t = e / (NT * 16 * MT * 16);
                z[((long)t * cop + m0 + row) * cipw + n0 + col] = 0.0f;
            }
        }
    }
}

template <int MT, int NT, int KD, int R, int XC>
int launch_wgrad_pers(const WgradArgs& a, int nblk, int cot, int cit, hipStream_t s) {
    constexpr int RB = R + 2;
    constexpr size_t lds = (size_t)(2 * (MT * R * XC * 4 + NT * KD * RB * (XC + 16) * 4) + 64) * 16;     // (the reduction reuses it)
    static_assert(lds <= 160 * 1024, "two unit buffers fit the LDS");
    auto kern = conv_wgrad_pers_kernel<MT, NT, KD, R, XC>;
    static unsigned long attr_done = 0;
    if (lds > 64 * 1024 && !mvconv::allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = mvconv::num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    const long x_bytes = (long)a.B * a.Di * a.Hi * a.Wi * a.CI * 4, gy_bytes = (long)a.B * a.Do * a.Ho * a.Wo * a.CO * 4;
    if (x_bytes >= (1L << 31) || gy_bytes >= (1L << 31)) return MVSTER_ERR_UNSUPPORTED;
    const int mgroups = cot / MT, ngroups = cit / NT;
    // workgroups that walk units: what fits the chip at once (LDS), spread over the channel-tile groups; the rest of the
    // nblk slots only write zeros
    int active = ncu / (mgroups * ngroups);                // one workgroup (8 waves, most of the LDS) per CU: wgrad_pers_slots
    if (active < 1) active = 1;
    if (active > nblk) active = nblk;
    MV_NOTE_KERNEL("conv_wgrad_pers_kernel<%d, %d, %d, %d, %d>", MT, NT, KD, R, XC);
    static const int dbg = MV_PROBE_ENV("MVSTER_WGRAD_DBG") ? atoi(MV_PROBE_ENV("MVSTER_WGRAD_DBG")) : 0;     // timing experiments only
    hipLaunchKernelGGL(kern, dim3(active, mgroups * ngroups), dim3(512), lds, s, a, mgroups, nblk, (unsigned)x_bytes, (unsigned)gy_bytes, dbg);
    return mv_check_launch();
}

const bool g_no_pers = MV_PROBE_ENV("MVSTER_WGRAD_NO_PERS") != nullptr;     // experiment switch: the staged kernels instead

}  // namespace

// slots the persistent kernel fills for this layer (= workgroups that walk units), 0 if it does not cover the layer
int wgrad_pers_slots(const WgradArgs& a, int cot, int cit) {
    if (g_no_pers || a.sd != 1 || a.sh != 1 || a.sw != 1 || a.kh != 3 || a.kw != 3 || a.ph != 1 || a.pw != 1 || (a.CO & 15) ||
        (a.CI & 15) || a.CO > 64 || a.CI > 64 || a.CO == 48 || a.CI == 48)     // (48: the callers round 3 tiles up to 4, the planes here hold 16 * tiles channels)
        return 0;
    if (!((a.kd == 1 && a.pd == 0) || (a.kd == 3 && a.pd == 1))) return 0;
    const int ncu = mvconv::num_cus();
    if (ncu <= 0) return 0;
    int mt = 1, nt = 1, per_cu = 1;
    if (a.kd == 1) {
        mt = cot >= 2 ? 2 : 1;
        nt = cit >= 2 ? 2 : 1;
    }
    const int groups = (cot / mt) * (cit / nt);
    int active = ncu * per_cu / groups;
    return active < 1 ? 1 : active;
}

int try_wgrad_pers(const WgradArgs& a, int nblk, int cot, int cit, hipStream_t s) {
    if (g_no_pers || a.sd != 1 || a.sh != 1 || a.sw != 1 || a.kh != 3 || a.kw != 3 || a.ph != 1 || a.pw != 1 || (a.CO & 15) ||
        (a.CI & 15) || a.CO > 64 || a.CI > 64 || a.CO == 48 || a.CI == 48)     // (48: the callers round 3 tiles up to 4, the planes here hold 16 * tiles channels)
        return MVSTER_ERR_UNSUPPORTED;
    if (!((a.kd == 1 && a.pd == 0) || (a.kd == 3 && a.pd == 1))) return MVSTER_ERR_UNSUPPORTED;
    // columns per unit: the multiple of 16 that leaves the fewest empty columns on this width (ties: the wider unit)
    const auto waste = [&](int xc) { return ((a.Wo + xc - 1) / xc) * xc - a.Wo; };
    if (a.kd == 1) {
        // 9 taps: up to 2 x 2 channel tiles (36 accumulator tiles) per workgroup; units of 64 or 80 columns
        const int mt = cot >= 2 ? 2 : 1, nt = cit >= 2 ? 2 : 1;
        const bool wide = waste(80) < waste(64);
        if (mt == 1 && nt == 1) return wide ? launch_wgrad_pers<1, 1, 1, 4, 80>(a, nblk, cot, cit, s) : launch_wgrad_pers<1, 1, 1, 4, 64>(a, nblk, cot, cit, s);
        if (mt == 2 && nt == 1) return wide ? launch_wgrad_pers<2, 1, 1, 4, 80>(a, nblk, cot, cit, s) : launch_wgrad_pers<2, 1, 1, 4, 64>(a, nblk, cot, cit, s);
        if (mt == 1 && nt == 2) return wide ? launch_wgrad_pers<1, 2, 1, 2, 80>(a, nblk, cot, cit, s) : launch_wgrad_pers<1, 2, 1, 4, 64>(a, nblk, cot, cit, s);
        return wide ? launch_wgrad_pers<2, 2, 1, 2, 80>(a, nblk, cot, cit, s) : launch_wgrad_pers<2, 2, 1, 2, 64>(a, nblk, cot, cit, s);
    }
    // 27 taps: one channel tile pair per workgroup; units of 64 or 48 columns (80 does not fit the LDS twice)
    if (waste(48) < waste(64)) return launch_wgrad_pers<1, 1, 3, 2, 48>(a, nblk, cot, cit, s);
    return launch_wgrad_pers<1, 1, 3, 2, 64>(a, nblk, cot, cit, s);
}

}  // namespace mvwgrad
