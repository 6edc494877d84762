// Weight gradient of the channels-last convolutions (training; autograd of the reference's
// nn.Conv3d / nn.Conv2d / nn.ConvTranspose3d layers, models/mvs4net_utils.py:116-123, :224-251,
// :870-965, :419-502).
//
//   dW[tap][co][ci] = sum over output voxels o of  gy[o][co] * x[o*s - p + tap][ci]
//
// is, per kernel tap, a [CO x P] x [P x CI] GEMM whose reduction dimension is the voxel count P
// (up to 2.6 M here) while CO and CI are 4..64.  MIOpen's solvers for these shapes take 30-50 ms
// per layer on gfx950; this kernel makes the voxels the K dimension of v_mfma_f32_16x16x4_f32:
//   lane (r = lane & 15, k = lane >> 4) of a K-step of four consecutive output columns loads
//   A = gy[x0 + k][co0 + r] and B = x[(x0 + k)*sw - pw + kx][ci0 + r], zero outside the image,
// one workgroup = one tap x a set of output rows (one row per wave at a time, so the row/tap
// bounds logic is wave-uniform and the inner loop only moves along x).  The four waves' partial
// tiles are summed through LDS and written to a per-workgroup slot of `partial`
// [nblk][taps][COT*16][CIT*16]; the host sums the slots (deterministic, no atomics).
#include <stdlib.h>

#include "conv_wgrad.hpp"

namespace {

using mvwgrad::WgradArgs;
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int COT, int CIT>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [3 waves][COT*CIT][64 lanes][4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, k = lane >> 4;
    const int tap = blockIdx.y, ntaps = gridDim.y;
    const int kx = tap % a.kw, ky = (tap / a.kw) % a.kh, kz = tap / (a.kw * a.kh);

    f32x4v acc[COT][CIT];
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    // channel columns of this lane, clamped (lanes beyond the channel count contribute zeros)
    int co[COT], ci[CIT];
    bool vco[COT], vci[CIT];
#pragma unroll
    for (int i = 0; i < COT; ++i) { co[i] = min(i * 16 + r, a.CO - 1); vco[i] = i * 16 + r < a.CO; }
#pragma unroll
    for (int j = 0; j < CIT; ++j) { ci[j] = min(j * 16 + r, a.CI - 1); vci[j] = j * 16 + r < a.CI; }

    const int nrows = a.B * a.Do * a.Ho;
    for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {    // wave-uniform
        const int yo = row % a.Ho, t = row / a.Ho;
        const int zo = t % a.Do, b = t / a.Do;
        const int iz = zo * a.sd - a.pd + kz, iy = yo * a.sh - a.ph + ky;
        if ((unsigned)iz >= (unsigned)a.Di || (unsigned)iy >= (unsigned)a.Hi) continue;   // this tap sees padding
        const float* grow = a.gy + (long)row * a.Wo * a.CO;
        const float* xrow = a.x + ((((long)b * a.Di + iz) * a.Hi + iy) * a.Wi) * a.CI;
        for (int x1 = 0; x1 < a.Wo; x1 += 16) {
            // four K-steps per trip: all their loads are issued before the first MFMA needs one
            float av[4][COT], bv[4][CIT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int xo = x1 + u * 4 + k;
                const int ix = xo * a.sw - a.pw + kx;
                const bool vo = xo < a.Wo;
                const bool vi = vo && (unsigned)ix < (unsigned)a.Wi;
                const int xoc = min(xo, a.Wo - 1);
                const int ixc = min(max(ix, 0), a.Wi - 1);
#pragma unroll
                for (int i = 0; i < COT; ++i) {
                    const float v = grow[xoc * a.CO + co[i]];
                    av[u][i] = (vo && vco[i]) ? v : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < CIT; ++j) {
                    const float v = xrow[ixc * a.CI + ci[j]];
                    bv[u][j] = (vi && vci[j]) ? v : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < COT; ++i)
#pragma unroll
                    for (int j = 0; j < CIT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
        }
    }

    // D fragment: lane holds rows 4*(lane>>4)+q (co), column lane&15 (ci).  Waves 1..3 hand their tiles to
    // wave 0 through LDS; wave 0 adds them in wave order and writes the workgroup's slot.
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int j = 0; j < CIT; ++j)
                *reinterpret_cast<f32x4v*>(&red[(((wave - 1) * COT * CIT + i * CIT + j) * 64 + lane) * 4]) = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + ((long)blockIdx.x * ntaps + tap) * (COT * 16) * (CIT * 16);
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
            for (int j = 0; j < CIT; ++j) {
                f32x4v s = acc[i][j];
#pragma unroll
                for (int w = 0; w < 3; ++w)
                    s += *reinterpret_cast<const f32x4v*>(&red[((w * COT * CIT + i * CIT + j) * 64 + lane) * 4]);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    out[(i * 16 + 4 * k + q) * (CIT * 16) + j * 16 + r] = s[q];
            }
    }
}

// Narrow input side (CI <= 8): a 16-wide N tile would be at least half padding, so TPN = 16 / CIP kernel taps share
// one tile instead: column n = (tap within the group) * CIP + ci.  Every lane then has its own tap, i.e. its own input
// row and column shift (no longer wave-uniform, but still one address computation per row and lane), and one MFMA
// does the work of TPN.  blockIdx.y = tap group; partial [nblk][groups][COT*16][16].
template <int COT, int CIP>
__global__ void __launch_bounds__(256) conv_wgrad_packed_kernel(WgradArgs a, int ntaps) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [3 waves][COT][64 lanes][4]
    constexpr int TPN = 16 / CIP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, k = lane >> 4;
    const int ngroups = gridDim.y;
    const int tap = blockIdx.y * TPN + r / CIP, ci = r % CIP;
    const bool vlane = tap < ntaps && ci < a.CI;
    const int tapc = min(tap, ntaps - 1), cic = min(ci, a.CI - 1);
    const int kx = tapc % a.kw, ky = (tapc / a.kw) % a.kh, kz = tapc / (a.kw * a.kh);

    f32x4v acc[COT];
#pragma unroll
    for (int i = 0; i < COT; ++i) acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    int co[COT];
    bool vco[COT];
#pragma unroll
    for (int i = 0; i < COT; ++i) { co[i] = min(i * 16 + r, a.CO - 1); vco[i] = i * 16 + r < a.CO; }

    const int nrows = a.B * a.Do * a.Ho;
    for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {    // wave-uniform
        const int yo = row % a.Ho, t = row / a.Ho;
        const int zo = t % a.Do, b = t / a.Do;
        const int iz = zo * a.sd - a.pd + kz, iy = yo * a.sh - a.ph + ky;         // per lane
        const bool vrow = vlane && (unsigned)iz < (unsigned)a.Di && (unsigned)iy < (unsigned)a.Hi;
        const int izc = min(max(iz, 0), a.Di - 1), iyc = min(max(iy, 0), a.Hi - 1);
        const float* grow = a.gy + (long)row * a.Wo * a.CO;
        const float* xrow = a.x + ((((long)b * a.Di + izc) * a.Hi + iyc) * a.Wi) * a.CI + cic;
        for (int x1 = 0; x1 < a.Wo; x1 += 16) {
            float av[4][COT], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int xo = x1 + u * 4 + k;
                const int ix = xo * a.sw - a.pw + kx;
                const bool vo = xo < a.Wo;
                const bool vi = vo && vrow && (unsigned)ix < (unsigned)a.Wi;
                const int xoc = min(xo, a.Wo - 1);
                const int ixc = min(max(ix, 0), a.Wi - 1);
#pragma unroll
                for (int i = 0; i < COT; ++i) {
                    const float v = grow[xoc * a.CO + co[i]];
                    av[u][i] = (vo && vco[i]) ? v : 0.0f;
                }
                const float v = xrow[ixc * a.CI];
                bv[u] = vi ? v : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < COT; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][i], bv[u], acc[i], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < COT; ++i)
            *reinterpret_cast<f32x4v*>(&red[(((wave - 1) * COT + i) * 64 + lane) * 4]) = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = a.partial + ((long)blockIdx.x * ngroups + blockIdx.y) * (COT * 16) * 16;
#pragma unroll
        for (int i = 0; i < COT; ++i) {
            f32x4v s = acc[i];
#pragma unroll
            for (int w = 0; w < 3; ++w) s += *reinterpret_cast<const f32x4v*>(&red[((w * COT + i) * 64 + lane) * 4]);
#pragma unroll
            for (int q = 0; q < 4; ++q) out[(i * 16 + 4 * k + q) * 16 + r] = s[q];
        }
    }
}

template <int COT, int CIP>
int launch_wgrad_packed(const WgradArgs& a, int nblk, int ntaps, hipStream_t s) {
    constexpr int TPN = 16 / CIP;
    const size_t lds = (size_t)3 * COT * 256 * sizeof(float);
    MV_NOTE_KERNEL("conv_wgrad_packed_kernel<%d, %d>", COT, CIP);
    hipLaunchKernelGGL((conv_wgrad_packed_kernel<COT, CIP>), dim3(nblk, (ntaps + TPN - 1) / TPN), dim3(256), lds, s, a, ntaps);
    return mv_check_launch();
}

template <int COT, int CIT>
int launch_wgrad(const WgradArgs& a, int nblk, int ntaps, hipStream_t s) {
    const size_t lds = (size_t)3 * COT * CIT * 256 * sizeof(float);
    MV_NOTE_KERNEL("conv_wgrad_kernel<%d, %d>", COT, CIT);
    hipLaunchKernelGGL((conv_wgrad_kernel<COT, CIT>), dim3(nblk, ntaps), dim3(256), lds, s, a);
    return mv_check_launch();
}

// ------------------------------------------------------------------------------------------
// LDS-staged form (what mvster_conv_wgrad runs whenever the layer fits it).  The kernels above feed every MFMA operand
// with its own 4-byte global load (a K step = four pixels x 16 channels = 256 B per wave instruction: a quarter of what
// the texture path moves per cycle with 16-byte loads) and re-read gy and x once per kernel tap; measured, the weight
// gradients of a training step run at 14 % of the fp32 MFMA peak.  Here one workgroup stages, per chunk of 64 output
// columns of one output row, the gy chunk and the kd*kh input rows it meets -- coalesced 16-byte loads, zero padding
// materialised -- and ALL kernel taps take their operands from LDS with conflict-free ds_read_b32 (pixel pitches chosen
// so that the four pixels of a K step land 16 banks apart).  A workgroup owns MT x NT channel tiles (blockIdx.y) and
// every tap: MT*NT*taps accumulator tiles live in registers across all rows the workgroup visits; the four waves split
// the K steps of a chunk and are summed through LDS once, at the end.  Chunks are register-double-buffered (the next
// chunk's global loads fly under this chunk's MFMAs).  Same `partial` layouts as above (host sum:
// deterministic).  PCB = 0: one tap per N tile; PCB = 4 / 8: narrow B side, 16 / PCB taps share an N tile.
// ------------------------------------------------------------------------------------------
constexpr int kXC = 64;      // output columns per staged chunk

__host__ __device__ constexpr int wg_pitch(int c, int s) {
    // floats per staged pixel: >= c, a multiple of 4, and s * pitch = 16 (mod 32) so that consecutive pixels of a K step
    // are 16 banks apart for ds_read_b32 (the two pixels a 32-lane group reads never collide)
    int p = (c + 3) & ~3;
    while ((s * p) % 32 != 16) p += 4;
    return p;
}

// Round 6: the x stride is a template parameter (1 or 2), so the staged-patch geometry (XB, PB) and with it every LDS
// offset of the MFMA phase is an instruction immediate, and everything about a staging slot that does not change from
// one chunk to the next (its global offset inside the unit, its LDS address, its patch row / column) is formed once per
// thread.  Before, the index arithmetic of the three staging lambdas and of the operand addresses -- runtime divisions by
// XB, the (row -> b, z, y) decode on the vector unit, a select per B operand -- was ~235 VALU instructions per wave and
// chunk beside 20 MFMAs on the narrow layers: 1 800 cycles per chunk and SIMD where the matrix pipe needs 640.
template <int MT, int NT, int TY, int KW, int PCB, int SW>
__global__ void __launch_bounds__(256) conv_wgrad_lds_kernel(WgradArgs a, int mgroups) {
    constexpr int TAPS = TY * KW;
    constexpr bool PACKED = PCB > 0;
    constexpr int TPN = PACKED ? 16 / PCB : 1;
    constexpr int NG = PACKED ? (TAPS + TPN - 1) / TPN : TAPS;      // N-tile groups per (mt, nt)
    constexpr int NTT = PACKED ? 1 : NT;
    constexpr int NACC = MT * NTT * NG;
    constexpr int PA = wg_pitch(MT * 16, 1);
    constexpr int CBB = PACKED ? PCB : NT * 16;                      // B channels staged per pixel
    constexpr int PB = wg_pitch(CBB, SW);
    constexpr int XB = (kXC - 1) * SW + KW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const As = lds;                                           // [kXC][PA]
    float* const Bs = lds + kXC * PA;                                // [TY][XB][PB]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, k = lane >> 4;
    const int m0 = (blockIdx.y % mgroups) * MT * 16, n0 = PACKED ? 0 : (blockIdx.y / mgroups) * NT * 16;

    f32x4v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};

    // MFMA operands of this lane: K step ks = wave + 4 it of a chunk = output columns ks*4 + k; the `it` part is an immediate.
    // A packed tile's lanes take different taps (r / PCB); a tap past the last one reads the last one's values -- its columns
    // of `partial` are dropped by the finish kernel.
    const float* const Ab = As + (wave * 4 + k) * PA + r;
    const float* const Bb = Bs + (wave * 4 + k) * SW * PB + (PACKED ? 0 : r);
    const float* bpk[PACKED ? NG : 1];
    if (PACKED) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int tap = min(g * TPN + r / (PACKED ? PCB : 1), TAPS - 1);
            const int ty = tap / KW, kx = tap - ty * KW;
            bpk[g] = Bb + (ty * XB + kx) * PB + r % (PACKED ? PCB : 1);
        }
    }

    const int nrows = a.B * a.Do * a.Ho;
    const int nchunks = (a.Wo + kXC - 1) / kXC;
    constexpr int qa = MT * 4, qb = CBB / 4;                         // float4 per staged pixel
    // A staging unit = one chunk of one output row.  Units are register-double-buffered: the global loads of unit u+1
    // are issued before the MFMAs of unit u and written to LDS after them (measured over the 64 weight gradients of a
    // config-4 step: 5.26 -> 4.40 ms; the 27-tap layers gain too, although the nine staged rows cost them 40 registers).
    constexpr int NA = (kXC * qa + 255) / 256;
    constexpr int nb4 = TY * XB * qb;                                // float4 of the B patch
    constexpr int NBX = (nb4 + 255) / 256;
    f32x4v ra[NA], rb[NBX];
    // per staging slot, once: byte offset inside the unit (from the unit's first gy column / first patch pixel), LDS
    // address, and for the bounds tests the chunk column (A; 255 = no such slot) or column | 1 << (8 + patch row) (B)
    unsigned a_g[NA], a_px[NA], b_g[NBX], b_pos[NBX];
    float* a_s[NA];
    float* b_s[NBX];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int px = idx / qa, q = idx - px * qa;
        const int c = m0 + q * 4;
        const bool ok = idx < kXC * qa && c < a.CO;
        a_g[i] = (unsigned)(px * a.CO + c) * 4u;
        a_px[i] = ok ? (unsigned)px : 255u;
        a_s[i] = As + px * PA + q * 4;
    }
#pragma unroll
    for (int i = 0; i < NBX; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int q = idx % qb, rest = idx / qb;
        const int px = rest % XB, ty = rest / XB;
        const int kz = ty / a.kh, ky = ty - kz * a.kh;
        const int c = n0 + q * 4;
        const bool ok = idx < nb4 && c < a.CI;
        b_g[i] = (unsigned)(((kz * a.Hi + ky) * a.Wi + px) * a.CI + c) * 4u;
        b_pos[i] = ok ? (unsigned)px | (1u << (8 + ty)) : 0u;
        b_s[i] = Bs + (ty * XB + px) * PB + q * 4;
    }
    // this workgroup's units: rows blockIdx.x, + gridDim.x, ...; all of it wave-uniform and stepped, not divided
    struct Unit {
        int chunk, b, zo, yo;
        long row;
    };
    Unit nu;
    nu.chunk = 0;
    nu.row = blockIdx.x;
    {
        const int yo = (int)(nu.row % a.Ho), t = (int)(nu.row / a.Ho);
        nu.yo = yo;
        nu.zo = t % a.Do;
        nu.b = t / a.Do;
    }
    auto advance = [&](Unit& u) {
        if (++u.chunk < nchunks) return;
        u.chunk = 0;
        u.row += gridDim.x;
        u.yo += (int)gridDim.x;
        while (u.yo >= a.Ho) {
            u.yo -= a.Ho;
            if (++u.zo == a.Do) {
                u.zo = 0;
                ++u.b;
            }
        }
    };
    auto fetch = [&](const Unit& u) {
        const int x1 = u.chunk * kXC;
        const int iz0 = u.zo * a.sd - a.pd, iy0 = u.yo * a.sh - a.ph, ix0 = x1 * SW - a.pw;
        // buffer descriptors at the unit's origin (the B origin may lie in front of the tensor: only lanes whose pixel is
        // inside the volume use it); 0xFFFFFFF0 is out of range -> zeros
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.gy + (u.row * a.Wo + x1) * a.CO), (short)0, (int)0xFFFFFF00u, 0x00020000);
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x + ((((long)u.b * a.Di + iz0) * a.Hi + iy0) * a.Wi + ix0) * a.CI), (short)0, (int)0xFFFFFF00u,
            0x00020000);
        const unsigned rem = (unsigned)min(a.Wo - x1, kXC);
        unsigned rowmask = 0;                                        // bit 8 + ty: patch row ty lies inside the volume
        int kz = 0, ky = 0;
#pragma unroll
        for (int ty = 0; ty < TY; ++ty) {
            if ((unsigned)(iz0 + kz) < (unsigned)a.Di && (unsigned)(iy0 + ky) < (unsigned)a.Hi) rowmask |= 1u << (8 + ty);
            if (++ky == a.kh) {
                ky = 0;
                ++kz;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned off = a_px[i] < rem ? a_g[i] : 0xFFFFFFF0u;
            ra[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(ars, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NBX; ++i) {
            const bool ok = (b_pos[i] & rowmask) != 0 && (unsigned)(ix0 + (int)(b_pos[i] & 255u)) < (unsigned)a.Wi;
            const unsigned off = ok ? b_g[i] : 0xFFFFFFF0u;
            rb[i] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(brs, off, 0, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if ((i + 1) * 256 <= kXC * qa || threadIdx.x + i * 256 < kXC * qa) *reinterpret_cast<f32x4v*>(a_s[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NBX; ++i)
            if ((i + 1) * 256 <= nb4 || threadIdx.x + i * 256 < nb4) *reinterpret_cast<f32x4v*>(b_s[i]) = rb[i];
    };
    auto compute = [&]() {
#pragma unroll
        for (int it = 0; it < kXC / 16; ++it) {
            float av[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = Ab[it * 16 * PA + i * 16];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int j = 0; j < NTT; ++j) {
                    float bv;
                    if (PACKED) {
                        bv = bpk[PACKED ? g : 0][it * 16 * SW * PB];
                    } else {
                        bv = Bb[it * 16 * SW * PB + ((g / KW) * XB + g % KW) * PB + j * 16];
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        acc[(i * NTT + j) * NG + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[(i * NTT + j) * NG + g], 0, 0, 0);
                }
            }
        }
    };
    const int my_rows = blockIdx.x < nrows ? (nrows - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int nunits = my_rows * nchunks;
    if (nunits > 0) {
        fetch(nu);
        advance(nu);
        commit();
    }
    __syncthreads();
    for (int u = 0; u < nunits; ++u) {
        if (u + 1 < nunits) {
            fetch(nu);
            advance(nu);
        }
        compute();
        __syncthreads();                                             // this unit's readers are done
        if (u + 1 < nunits) commit();
        __syncthreads();
    }

    // cross-wave sum, one accumulator tile at a time through LDS (3 KB), then the workgroup's slot of `partial`
    const int cop = (((a.CO + 15) / 16 == 3) ? 4 : (a.CO + 15) / 16) * 16;
    const int cipw = PACKED ? 16 : (((a.CI + 15) / 16 == 3) ? 4 : (a.CI + 15) / 16) * 16;
    float* slot = a.partial + (long)blockIdx.x * NG * cop * cipw;        // [NG][cop][cipw]
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTT; ++j)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const f32x4v mine = acc[(i * NTT + j) * NG + g];
                if (wave > 0) *reinterpret_cast<f32x4v*>(lds + ((wave - 1) * 64 + lane) * 4) = mine;
                __syncthreads();
                if (wave == 0) {
                    f32x4v sum = mine;
#pragma unroll
                    for (int w = 0; w < 3; ++w) sum += *reinterpret_cast<const f32x4v*>(lds + (w * 64 + lane) * 4);
                    float* out = slot + (long)g * cop * cipw;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mrow = m0 + i * 16 + 4 * k + q, ncol = n0 + j * 16 + r;
                        if (mrow < cop && ncol < cipw) out[mrow * cipw + ncol] = sum[q];
                    }
                }
                __syncthreads();
            }
}

template <int MT, int NT, int TY, int KW, int PCB, int SW>
int launch_wgrad_lds_sw(const WgradArgs& a, int nblk, int cot, int cit, hipStream_t s) {
    constexpr int PA = wg_pitch(MT * 16, 1), PB = wg_pitch(PCB > 0 ? PCB : NT * 16, SW);
    constexpr int XB = (kXC - 1) * SW + KW;
    constexpr size_t lds = sizeof(float) * ((size_t)kXC * PA + (size_t)TY * XB * PB);
    // (layers that need more than 64 KB -- stride-2 3x3 from 32 channels, 3x3 from 64 -- were measured no faster here with the
    //  limit raised to 128 KB, one workgroup per CU, than on the per-tap kernels below: 105 vs 47+ us, 642 vs 588 us.)
    // The 5x5 stride-2 layers of the FPN (16 -> 32, 32 -> 64) need 65.4 KB: two workgroups per CU still fit the 160 KB.
    static const bool big5 = MV_PROBE_ENV("MVSTER_WGRAD_NO_BIG5") == nullptr;
    const size_t limit = (TY == 5 && PCB == 0 && big5) ? 80 * 1024 : 64 * 1024;
    if (lds > limit) return MVSTER_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        static unsigned long allowed = 0;                          // per kernel and per device ordinal
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MVSTER_ERR_UNSUPPORTED;
        if (!((allowed >> dev) & 1ul)) {
            if (hipFuncSetAttribute((const void*)conv_wgrad_lds_kernel<MT, NT, TY, KW, PCB, SW>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess)
                return MVSTER_ERR_UNSUPPORTED;
            allowed |= 1ul << dev;
        }
    }
    const int mgroups = cot / MT, ngroups = PCB > 0 ? 1 : cit / NT;
    MV_NOTE_KERNEL("conv_wgrad_lds_kernel<%d, %d, %d, %d, %d, %d>", MT, NT, TY, KW, PCB, SW);
    hipLaunchKernelGGL((conv_wgrad_lds_kernel<MT, NT, TY, KW, PCB, SW>), dim3(nblk, mgroups * ngroups), dim3(256), lds, s, a, mgroups);
    return mv_check_launch();
}

// x stride 1, and 2 for the 3x3 / 5x5 layers (the only strided ones of the network)
template <int MT, int NT, int TY, int KW, int PCB>
int launch_wgrad_lds(const WgradArgs& a, int nblk, int cot, int cit, hipStream_t s) {
    if (a.sw == 1) return launch_wgrad_lds_sw<MT, NT, TY, KW, PCB, 1>(a, nblk, cot, cit, s);
    if constexpr ((TY == 3 || TY == 5) && KW == TY) {
        if (a.sw == 2) return launch_wgrad_lds_sw<MT, NT, TY, KW, PCB, 2>(a, nblk, cot, cit, s);
    }
    return MVSTER_ERR_UNSUPPORTED;
}

// Accumulator tiles (MT * NT * tap groups) a wavefront may hold.  Small tiles win: the staging of a chunk is not
// overlapped with its own math, so it is occupancy that hides it -- measured over the 64 weight gradients of a config-4
// step: limit 40 -> 6.09 ms, 20 -> 5.49 ms, 12 -> 5.28 ms, 4 -> 5.22 ms.  MVSTER_WGRAD_ACC overrides (experiments).
static int wgrad_acc_limit() {
    static const int v = MV_PROBE_ENV("MVSTER_WGRAD_ACC") ? atoi(MV_PROBE_ENV("MVSTER_WGRAD_ACC")) : 12;
    return v;
}

// tile shapes per tap count: MT*NT*taps accumulator tiles of 4 registers must stay well below the register file
template <int TY, int KW>
int dispatch_wgrad_lds(const WgradArgs& a, int nblk, int cot, int cit, int packed, hipStream_t s) {
    constexpr int TAPS = TY * KW;
    if (packed) {
        const int pcb = a.CI <= 4 ? 4 : 8;
        if (TAPS == 1) return MVSTER_ERR_UNSUPPORTED;
#define MV_LP(M_, P_) if (mt == M_ && pcb == P_) return launch_wgrad_lds<M_, 1, TY, KW, P_>(a, nblk, cot, cit, s);
        const int ng8 = (TAPS + 1) / 2, ng4 = (TAPS + 3) / 4;
        int mt = cot;
        while (mt > 1 && mt * (pcb == 8 ? ng8 : ng4) > wgrad_acc_limit()) mt /= 2;
        MV_LP(1, 4) MV_LP(2, 4) MV_LP(4, 4) MV_LP(1, 8) MV_LP(2, 8) MV_LP(4, 8)
#undef MV_LP
        return MVSTER_ERR_UNSUPPORTED;
    }
    int mt = cot, nt = cit;
    if (cot == 5 && TAPS != 1) return MVSTER_ERR_UNSUPPORTED;        // 5 M tiles do not halve
    while (mt * nt * TAPS > wgrad_acc_limit() && (mt > 1 || nt > 1)) {
        if (mt >= nt && mt > 1 && mt != 5) mt /= 2;
        else if (nt > 1) nt /= 2;
        else break;
    }
    if (mt * nt * TAPS > 40) return MVSTER_ERR_UNSUPPORTED;
#define MV_LN(M_, N_) if (mt == M_ && nt == N_) { if constexpr (M_ * N_ * TAPS <= 40) return launch_wgrad_lds<M_, N_, TY, KW, 0>(a, nblk, cot, cit, s); }
    MV_LN(1, 1) MV_LN(2, 1) MV_LN(1, 2) MV_LN(2, 2) MV_LN(4, 1) MV_LN(1, 4) MV_LN(4, 2) MV_LN(2, 4) MV_LN(4, 4)
    if constexpr (TAPS == 1) { MV_LN(5, 4) MV_LN(5, 2) MV_LN(5, 1) }      // 72 (+8) x 64: the FPN gather's 1x1 conv
#undef MV_LN
    return MVSTER_ERR_UNSUPPORTED;
}

static const bool g_wgrad_no_lds = MV_PROBE_ENV("MVSTER_WGRAD_NO_LDS") != nullptr;   // experiment switch: the per-tap kernels

int try_wgrad_lds(const WgradArgs& a, int nblk, int cot, int cit, int packed, hipStream_t s) {
    if (g_wgrad_no_lds || (a.CO & 3) || (a.CI & 3)) return MVSTER_ERR_UNSUPPORTED;
    const int ty = a.kd * a.kh;
    if (ty == 1 && a.kw == 1) return dispatch_wgrad_lds<1, 1>(a, nblk, cot, cit, packed, s);
    if (ty == 3 && a.kw == 3) return dispatch_wgrad_lds<3, 3>(a, nblk, cot, cit, packed, s);
    if (ty == 9 && a.kw == 3) return dispatch_wgrad_lds<9, 3>(a, nblk, cot, cit, packed, s);
    if (ty == 5 && a.kw == 5) return dispatch_wgrad_lds<5, 5>(a, nblk, cot, cit, packed, s);
    return MVSTER_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// Finish: add the nblk slots of `partial` (fixed order: deterministic) and write the gradient in the parameter's own
// layout -- one launch instead of a tensor reduction plus a permuting copy (plus flips / transposes for the mirrored
// form).  A workgroup owns 64 consecutive elements of a slot; its 16 wavefronts stride over the slots (every load is a
// 256-byte run), meet in LDS, and the first wavefront scatters the 64 sums.
//   element e = (g, row, col) of a slot [ngrp][cop][width]:
//     co = row;  not packed: tap = g, ci = col;  packed (cip = 4 / 8): tap = g * (16 / cip) + col / cip, ci = col % cip
//   kept when tap < ntaps, co < co_lim, ci < ci_lim; flip: tap -> ntaps - 1 - tap;
//   dw[co][ci][tap] ([co_lim][ci_lim][ntaps]), or with swap dw[ci][co][tap] ([ci_lim][co_lim][ntaps]).
// ------------------------------------------------------------------------------------------
struct FinishArgs {
    const float* partial; float* dw;
    int nblk, ngrp, cop, width, ntaps, cip, co_lim, ci_lim, swap, flip;
};

__device__ __forceinline__ void finish_block(const FinishArgs& a, long block) {
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const long E = (long)a.ngrp * a.cop * a.width;
    const long e = block * 64 + el;
    float s = 0.0f;
    if (e < E) {
        // eight slots in flight per thread (the loads are issued clamped and masked afterwards: one dependent load after
        // the other made this kernel a chain of L2 latencies, 7 us per launch and 64 launches per training step)
        const float* __restrict__ p = a.partial + e;
        for (int n0 = lane; n0 < a.nblk; n0 += 128) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + 16 * j;
                v[j] = p[(long)(n < a.nblk ? n : n0) * E];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = n0 + 16 * j < a.nblk ? v[j] : 0.0f;
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
    }
    red[lane][el] = s;
    __syncthreads();
    if (lane != 0 || e >= E) return;
#pragma unroll
    for (int w = 1; w < 16; ++w) s += red[w][el];
    const int col = (int)(e % a.width), row = (int)((e / a.width) % a.cop), g = (int)(e / ((long)a.width * a.cop));
    int tap = g, ci = col;
    if (a.cip) { tap = g * (16 / a.cip) + col / a.cip; ci = col % a.cip; }
    if (tap >= a.ntaps || row >= a.co_lim || ci >= a.ci_lim) return;
    if (a.flip) tap = a.ntaps - 1 - tap;
    const long o = a.swap ? ((long)ci * a.co_lim + row) * a.ntaps + tap : ((long)row * a.ci_lim + ci) * a.ntaps + tap;
    a.dw[o] = s;
}

__global__ void __launch_bounds__(1024) conv_wgrad_finish_kernel(FinishArgs a) { finish_block(a, blockIdx.x); }

// Every pending finish of a training step in one launch: the weight gradients are read by nobody before the backward
// pass is over (no hook-driven reducer: see train_ops.deferred_wgrad_finish), so the 64 finishing launches of 5 us become
// one or two.  The records travel as kernel arguments (the partial buffers are fresh allocations every step).
constexpr int kFinishBatch = 56;       // (56-byte records + a 4-byte block offset each: the argument block stays under 4 KB)
struct FinishBatchArgs {
    FinishArgs a[kFinishBatch];
    int first_block[kFinishBatch];
    int count;
};

__global__ void __launch_bounds__(1024) conv_wgrad_finish_batch_kernel(FinishBatchArgs b) {
    int lo = 0, hi = b.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.first_block[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    finish_block(b.a[lo], (long)((int)blockIdx.x - b.first_block[lo]));
}

}  // namespace

// x [B,Di,Hi,Wi,CI], gy [B,Do,Ho,Wo,CO] (channels-last, contiguous); partial [nblk][kd*kh*kw][COP][CIP] with
// COP / CIP = CO / CI rounded up to 16 (CI <= 64, CO <= 80).  Do/Ho/Wo must be the conv's output size for (k, s, p).
// packed = 0: partial [nblk][taps][COP][CIP16] (one kernel tap per N tile).
// packed = 1 (CI <= 8): partial [nblk][ceil(taps/TPN)][COP][16] with TPN = 16/CIP taps per tile, CIP = 4 or 8 (CI rounded
// up); column n of a tile = (tap % TPN) * CIP + ci.
extern "C" int mvster_conv_wgrad(const float* x, const float* gy, float* partial, int nblk, int B, int Di, int Hi, int Wi,
                                 int CI, int Do, int Ho, int Wo, int CO, int kd, int kh, int kw, int sd, int sh, int sw,
                                 int pd, int ph, int pw, int packed, void* stream) {
    if (!x || !gy || !partial) return MVSTER_ERR_NULL;
    if (B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || nblk <= 0 || kd <= 0 || kh <= 0 ||
        kw <= 0 || sd <= 0 || sh <= 0 || sw <= 0 || pd < 0 || ph < 0 || pw < 0)
        return MVSTER_ERR_SHAPE;
    if (CI <= 0 || CI > 64 || CO <= 0 || CO > 80) return MVSTER_ERR_UNSUPPORTED;
    if (Do != (Di + 2 * pd - kd) / sd + 1 || Ho != (Hi + 2 * ph - kh) / sh + 1 || Wo != (Wi + 2 * pw - kw) / sw + 1)
        return MVSTER_ERR_SHAPE;
    if ((long)Wo * CO >= (1L << 31) || (long)Wi * CI >= (1L << 31) || (long)kd * kh * kw > 65535) return MVSTER_ERR_SHAPE;
    WgradArgs a;
    a.x = x; a.gy = gy; a.partial = partial;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.CI = CI; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.CO = CO;
    a.kd = kd; a.kh = kh; a.kw = kw; a.sd = sd; a.sh = sh; a.sw = sw; a.pd = pd; a.ph = ph; a.pw = pw;
    const int cot = (CO + 15) / 16 == 3 ? 4 : (CO + 15) / 16, cit = (CI + 15) / 16 == 3 ? 4 : (CI + 15) / 16;
    const int ntaps = kd * kh * kw;
    hipStream_t s = (hipStream_t)stream;
    if (packed && CI > 8) return MVSTER_ERR_UNSUPPORTED;
    if (!packed) {
        const int rc = mvwgrad::try_wgrad_pers(a, nblk, cot, cit, s);
        if (rc != MVSTER_ERR_UNSUPPORTED) return rc;
    }
    {
        const int rc = try_wgrad_lds(a, nblk, cot, cit, packed, s);
        if (rc != MVSTER_ERR_UNSUPPORTED) return rc;
    }
    if (packed) {
        const int cip = CI <= 4 ? 4 : 8;
#define MV_P(A_, B_) if (cot == A_ && cip == B_) return launch_wgrad_packed<A_, B_>(a, nblk, ntaps, s);
        MV_P(1, 4) MV_P(1, 8) MV_P(2, 4) MV_P(2, 8) MV_P(4, 4) MV_P(4, 8)
#undef MV_P
        return MVSTER_ERR_UNSUPPORTED;
    }
#define MV_W(A_, B_) if (cot == A_ && cit == B_) return launch_wgrad<A_, B_>(a, nblk, ntaps, s);
    MV_W(1, 1) MV_W(1, 2) MV_W(1, 4) MV_W(2, 1) MV_W(2, 2) MV_W(2, 4) MV_W(4, 1) MV_W(4, 2) MV_W(4, 4) MV_W(5, 4)
#undef MV_W
    return MVSTER_ERR_UNSUPPORTED;
}

// Slot count the caller should give `partial` for this layer: the persistent kernel's workgroup count where it applies
// (more slots would only be zero-filled and re-read by the finish), else 0 = the caller's own rule.
extern "C" int mvster_conv_wgrad_slots(int CI, int CO, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                                       int packed) {
    if (CI <= 0 || CO <= 0) return 0;
    // The 5x5 stride-2 layers of the FPN on the LDS kernel: one resident round of workgroups (two per CU; measured 8 -> 16
    // 108 -> 99 us at 512 slots, 16 -> 32 127 -> 102 us at 256, 32 -> 64 145 -> 130 us at 64 -- the tail round of a
    // larger grid costs more than the parallelism gives).
    if (kd == 1 && kh == 5 && kw == 5 && sw == 2 && sh == 2 && !(CI & 3) && !(CO & 3)) {
        const int groups = packed ? 1 : ((CO + 15) / 16) * ((CI + 15) / 16);
        if (groups <= 8) return 512 / groups;
    }
    if (packed) return 0;
    WgradArgs a{};
    a.CI = CI; a.CO = CO; a.kd = kd; a.kh = kh; a.kw = kw; a.sd = sd; a.sh = sh; a.sw = sw; a.pd = pd; a.ph = ph; a.pw = pw;
    const int cot = (CO + 15) / 16 == 3 ? 4 : (CO + 15) / 16, cit = (CI + 15) / 16 == 3 ? 4 : (CI + 15) / 16;
    return mvwgrad::wgrad_pers_slots(a, cot, cit);
}

// partial [nblk][ngrp][cop][width] as written by mvster_conv_wgrad -> dw in parameter layout (see the kernel): ntaps =
// kd*kh*kw, cip = 0 (one tap per group) or 4 / 8 (packed), co_lim <= cop and ci_lim = the channel counts to keep.
extern "C" int mvster_conv_wgrad_finish(const float* partial, float* dw, int nblk, int ngrp, int cop, int width, int ntaps,
                                        int cip, int co_lim, int ci_lim, int swap, int flip, void* stream) {
    if (!partial || !dw) return MVSTER_ERR_NULL;
    if (nblk <= 0 || ngrp <= 0 || cop <= 0 || width <= 0 || ntaps <= 0 || co_lim <= 0 || ci_lim <= 0 || co_lim > cop)
        return MVSTER_ERR_SHAPE;
    if (cip != 0 && cip != 4 && cip != 8) return MVSTER_ERR_UNSUPPORTED;
    if (cip ? (width != 16 || ci_lim > cip || (long)ngrp * (16 / cip) < ntaps) : (ci_lim > width || ngrp != ntaps))
        return MVSTER_ERR_SHAPE;
    FinishArgs a{partial, dw, nblk, ngrp, cop, width, ntaps, cip, co_lim, ci_lim, swap, flip};
    const long E = (long)ngrp * cop * width;
    hipLaunchKernelGGL(conv_wgrad_finish_kernel, dim3((unsigned)((E + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

// `count` finishes in ceil(count / 56) launches: recs = HOST array of 56-byte records {const float* partial; float* dw; int
// nblk, ngrp, cop, width, ntaps, cip, co_lim, ci_lim, swap, flip} (the arguments of mvster_conv_wgrad_finish, same checks).
extern "C" int mvster_conv_wgrad_finish_batch(const void* recs, int count, void* stream) {
    if (!recs) return MVSTER_ERR_NULL;
    if (count <= 0) return MVSTER_ERR_SHAPE;
    static_assert(sizeof(FinishArgs) == 56, "FinishArgs layout");
    const FinishArgs* r = static_cast<const FinishArgs*>(recs);
    for (int i = 0; i < count; ++i) {
        const FinishArgs& a = r[i];
        if (!a.partial || !a.dw) return MVSTER_ERR_NULL;
        if (a.nblk <= 0 || a.ngrp <= 0 || a.cop <= 0 || a.width <= 0 || a.ntaps <= 0 || a.co_lim <= 0 || a.ci_lim <= 0 ||
            a.co_lim > a.cop)
            return MVSTER_ERR_SHAPE;
        if (a.cip != 0 && a.cip != 4 && a.cip != 8) return MVSTER_ERR_UNSUPPORTED;
        if (a.cip ? (a.width != 16 || a.ci_lim > a.cip || (long)a.ngrp * (16 / a.cip) < a.ntaps)
                  : (a.ci_lim > a.width || a.ngrp != a.ntaps))
            return MVSTER_ERR_SHAPE;
    }
    for (int i0 = 0; i0 < count; i0 += kFinishBatch) {
        FinishBatchArgs b;
        b.count = 0;
        long blocks = 0;
        for (int i = i0; i < count && b.count < kFinishBatch; ++i) {
            b.a[b.count] = r[i];
            b.first_block[b.count] = (int)blocks;
            blocks += ((long)r[i].ngrp * r[i].cop * r[i].width + 63) / 64;
            ++b.count;
        }
        hipLaunchKernelGGL(conv_wgrad_finish_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, b);
    }
    return mv_check_launch();
}
