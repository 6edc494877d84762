// 3x3 (and 3x3x3) stride-1 convolutions of the 16 / 32 / 64-channel layers with fp32 PRODUCTS ON THE BF16 MATRIX CORES
// (round 5 probe, variant 11 of mvster_conv_mfma).
//
// Why: on gfx950 the fp32-input MFMA runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD, 157 TFLOP/s): the dense layers of the
// forward (78 GFLOP) cannot take less than ~0.5 ms on it, and the Winograd kernels that carry most of them sit at 0.23-0.29
// of that peak because their transforms share the same VALU budget.  The bf16 MFMA is 16x faster.  An fp32 number splits
// EXACTLY into three bf16 numbers (24 significand bits = 8 + 8 + 8, round-to-nearest at every level, residuals computed
// exactly in fp32): x = x1 + x2 + x3.  A product a*b then is the sum of nine bf16 x bf16 products, each exact in fp32; the
// three smallest (a2 b3, a3 b2, a3 b3 <= 2^-24 |ab|, random sign) are dropped, the other six run as six bf16 MFMAs with fp32
// accumulation: 6/16 = 0.375 of the fp32-MFMA time at the accuracy of an fp32 dot product (dropped terms ~2^-24 relative
// per product, i.e. one fp32 rounding; the terms accumulate in three registers by magnitude -- 1, 2^-8, 2^-16 -- so the
// corrections are not rounded at the magnitude of the running sum).
//
// Frame (v2, persistent): workgroup = 512 threads = 4 compute waves + 4 loading waves, alive over its share of the work items
// (item = TY x 32 output pixels x NW = 16 NTW output channels of one (b, z) slice; TY = 4 TYQ rows, compute wave w owns TYQ
// rows).  The K loop of an item runs over stages = (kd, 16 input channels); the stages of all items of a workgroup form ONE
// stream that the loading waves run ahead of the compute waves: global loads of stage k + 2 in flight (registers), stage
// k + 1 being split into three bf16 planes and written to the other LDS buffer, stage k under the MFMAs -- one barrier
// per stage.  (v1, one tile per workgroup and everything in sequence, measured 1.0x the Winograd
// kernels: a wave was alive for ~6 us around 0.4 us of matrix work, profiles/r05_b3_v1_check.txt.)
// A stage in LDS: the input patch ((TY + 2) x 34 pixels x 16 channels) as [plane][8-channel half][row][column] x 16 B, so a
// 16-lane fragment read is 256 contiguous bytes, and -- unless the layer's weights live in registers (WREG: the one-stage
// 16 -> 16 layers, 60 registers) -- the stage's weights, which arrive pre-split in fragment order (conv_plan.py:pack_b3).
// MFMA shape: v_mfma_f32_16x16x32_bf16 with A = weights (M = 16 output channels), B = pixels (N = 16 pixels of a row),
// K = 32 = two taps x 16 channels (the nine taps of a slice pair up as (0,1) (2,3) (4,5) (6,7) (8, zero)); lane (p = lane &
// 15, g = lane >> 4) holds 8 consecutive channels (half g & 1) of tap 2 tp + (g >> 1) -- the same rule on both operands, so
// no K permutation exists.  D layout: column = lane & 15 = pixel, row = 4 g + reg = output channel: each lane ends up with
// four consecutive output channels of one pixel = one 16-byte store.
// Epilogue as in every other kernel of the family: y = acc * scale[co] + shift[co], ReLU, + skip (same resolution).
// Reference layers: models/mvs4net_utils.py:430-446 (FPN conv1-3 3x3 layers), :457-459 (out2 / out3), :877-883 (reg2d
// conv2 / conv4 / conv6).
//
// STATUS (round 5): a PROBE, compiled into libmvster_hip_probes.so only (make probes) and never selected by the plan.
// Accuracy gate passed: max error against fp64 0.44x the direct fp32 kernel's and equal to the Winograd kernels' on every
// layer shape of the forward (2.0e-7 .. 5.4e-7 of max |y|, profiles/r05_b3_v3_check.txt).  Speed gate not passed: 0.83x ..
// 1.22x the Winograd kernels (255 us against 245 us over the 14 layers of >= 20 000 voxels).  The s_memtime timeline
// (profiles/r05_b3_v3_timeline.txt) shows why the 0.375x matrix time does not come through:
//   * a SIMD issues matrix and vector instructions through one port: beside a stream of MFMAs a vector instruction gets
//     through about once per MFMA, and every one of them costs the compute wave of that SIMD ~8 cycles of matrix pipe.
//     The split needs ~90 vector instructions per lane and stage (5.5 per fp32 element) beside 120 MFMAs: the matrix pipe
//     runs at 25-27 cycles per MFMA instead of 16.5, and the loading wave's split takes 1 500-1 900 cycles;
//   * LDS-DMA costs its wave ~100 cycles of issue per kilobyte (the CU's 64 B/clk vector-memory path): 30 KB of pre-split
//     weights + 16 KB of fp32 patch per stage are 1 300 cycles per loading wave, 2/3 of it weights that a 4-row tile
//     re-fetches per stage (8-row tiles do not fit beside the raw ring: 202 KB);
//   * so a stage takes ~4 500 cycles around 1 980 cycles of matrix work -- the same 2.3x that the Winograd kernels lose
//     to their transforms.  What would change it is a producer that writes its output already split (no VALU work and no
//     raw ring here, 8-row tiles, half the weight traffic): an inter-layer format change, not a kernel.
#include "conv_args.hpp"

#ifdef MVSTER_PROBES
namespace {

using mvconv::ConvArgs;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

// Probe build only (make probes): s_memtime stamps of the first workgroups' waves at the phase boundaries of every stage
// (scripts/conv_b3_timeline.py).  Record = 4 x u64 per (workgroup < 4, wave 0..7, stage < 32): compute waves [0] past the
// barrier, [1] MFMAs issued, [2] epilogue stores issued (item end); loading waves [0] loop top, [1] split + LDS writes
// issued, [2] next stage's loads issued, [3] past the barrier.
#ifdef MVSTER_PROBES
__device__ unsigned long long* g_b3tl = nullptr;
#define B3_TL(slot)                                                                                                     \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (g_b3tl && lane == 0 && blockIdx.x < 4 && k < 32)                                                            \
            g_b3tl[((blockIdx.x * 8 + wave8) * 32 + k) * 4 + (slot)] = __builtin_amdgcn_s_memtime();                     \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define B3_TL(slot)
#endif

struct B3Args {
    const float* in;      // [B, D, H, W, CIN]
    const bf16x8* w;      // [KD][CIN/16][nsplit][1024 * NTW] 16-byte units (fragment order, zero-padded to 16 KB per N tile)
    const float* scale;   // [COUT]
    const float* shift;   // [COUT]
    const float* skip;    // [B, D, H, W, COUT] or null
    float* out;           // [B, D, H, W, COUT]
    int B, D, H, W, cout, relu;
    unsigned tiles_x, tiles_y, nsplit, ntiles;
    int dbg;              // probe runs: 1 = loading waves only pass the barriers, 2 = compute waves skip the MFMA loop
    int prio;             // 0: no priorities, 1: compute waves raised, 2: loading waves raised (experiment switch)
    unsigned w_bytes;     // size of the packed weights
    unsigned in_bytes;    // size of `in` (< 4 GB): range of the buffer descriptor the loading waves read through
};

template <int TYQ>
struct B3Geom {
    static constexpr int TY = 4 * TYQ, PH = TY + 2, PW = 34;
    // one 8-channel half of a plane, padded to a multiple of 256 B: ds_read_b128 serves the lanes of fragment groups g and
    // g + 1 (the two halves) in the same LDS cycle, conflict-free only if the halves are congruent modulo the 256-byte bank
    // row (MI355X_MICROARCH.md, LDS; measured 26-44 % conflict cycles with the unpadded 5 440-byte stride)
    static constexpr int HALF = (PH * PW + 15) / 16 * 16;
    static constexpr int PLANE = 2 * HALF;                    // 16-byte units of one bf16 plane (two 8-channel halves)
    static constexpr int PATCH = 3 * PLANE;
    static constexpr int PIX = PH * PW;
    // loading thread u handles pixel (u & 7) | ((u >> 4) << 3), half (u >> 3) & 1: eight consecutive lanes write eight
    // consecutive 16-byte slots (ds_write_b128 is served in contiguous 8-lane groups)
    // raw fp32 stage in LDS (target of the LDS-DMA): [pixel][4-channel quarter] x 16 B, pixels padded to whole DMA
    // instructions (one instruction = 16 pixels x 64 contiguous bytes) and to a whole number of them per loading wave
    static constexpr int PIXP = (PIX + 63) / 64 * 64;
    static constexpr int RAWU = 4 * PIXP;                     // 16-byte units of one raw stage
    static constexpr int NRIW = PIXP / 64;                    // DMA instructions per loading wave and stage (patch)
    static constexpr int NSU = RAWU / 256;                    // raw units per loading thread (split pass)
};

// x -> (bf16(x), x - bf16(x)): round-to-nearest-even conversion, exact fp32 residual
__device__ __forceinline__ void split_pair(float a, float b, bf16x2& p, float& ra, float& rb) {
    p[0] = (__bf16)a;
    p[1] = (__bf16)b;
    ra = a - (float)p[0];
    rb = b - (float)p[1];
}

__device__ __forceinline__ void split8(const f32x4b lo, const f32x4b hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
    float e[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x2 a, b, c;
        float r0, r1, s0, s1, t0, t1;
        split_pair(e[2 * i], e[2 * i + 1], a, r0, r1);
        split_pair(r0, r1, b, s0, s1);
        split_pair(s0, s1, c, t0, t1);
        p1[2 * i] = a[0]; p1[2 * i + 1] = a[1];
        p2[2 * i] = b[0]; p2[2 * i + 1] = b[1];
        p3[2 * i] = c[0]; p3[2 * i + 1] = c[1];
    }
}

struct B3Item { int ns, b, z, y0, x0, s_begin, s_end; };

__device__ __forceinline__ void split4(const f32x4b v, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        bf16x2 a, b, c;
        float r0, r1, s0, s1, t0, t1;
        split_pair(v[2 * i], v[2 * i + 1], a, r0, r1);
        split_pair(r0, r1, b, s0, s1);
        split_pair(s0, s1, c, t0, t1);
        p1[2 * i] = a[0]; p1[2 * i + 1] = a[1];
        p2[2 * i] = b[0]; p2[2 * i + 1] = b[1];
        p3[2 * i] = c[0]; p3[2 * i + 1] = c[1];
    }
}

template <int CIN, int NTW, int KD, int TYQ, bool WREG>
__global__ void __launch_bounds__(512) conv_b3_kernel(B3Args a) {
    using G = B3Geom<TYQ>;
    constexpr int PW = G::PW, PLANE = G::PLANE, NCH = CIN / 16;
    constexpr int PT = 2 * TYQ;                                // 16-pixel tiles per compute wave
    constexpr int GP = 2 / NTW, UPT = PT / GP, NUNIT = 5 * UPT;   // pixel tiles per pipeline unit, units per tap pair and stage
    constexpr int WUNITS = 1024 * NTW;                         // 16-byte units of a stage's weight block (padded)
    constexpr int NWI = WREG ? 0 : WUNITS / 256;               // weight DMA instructions per loading wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16x8* const lds = reinterpret_cast<bf16x8*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const unsigned nwg = gridDim.x, nitems = a.ntiles * a.nsplit;
    const unsigned first = xcd_remap(blockIdx.x, nwg);
    if (first >= nitems) return;

    auto decode = [&](unsigned item) {
        B3Item it;
        it.ns = (int)(item % a.nsplit);
        unsigned t = item / a.nsplit;
        const int tx = (int)(t % a.tiles_x); t /= a.tiles_x;
        const int ty = (int)(t % a.tiles_y); t /= a.tiles_y;
        it.z = (int)(t % (unsigned)a.D);
        it.b = (int)(t / (unsigned)a.D);
        it.y0 = ty * G::TY;
        it.x0 = tx * 32;
        // stages: kd in [kd_lo, kd_hi] (the slices inside the volume) x the NCH channel chunks
        const int kd_lo = KD == 1 ? 0 : (it.z == 0 ? 1 : 0);
        const int kd_hi = KD == 1 ? 0 : (it.z == a.D - 1 ? KD - 2 : KD - 1);
        it.s_begin = kd_lo * NCH;
        it.s_end = (kd_hi + 1) * NCH;
        return it;
    };

    // LDS: [raw ring 3 x RAWU][bf16 planes 2 x PATCH][weight ring 2 x WUNITS]
    bf16x8* const raw_ring = lds;
    bf16x8* const planes = lds + 3 * G::RAWU;
    bf16x8* const w_ring = planes + 2 * G::PATCH;

    if (loader) {
        // ------------------------------------------------------------------------------------------------ loading waves
        // Per stage j (between barrier j - 1 and barrier j):  (1) LDS-DMA of stage j's weights into weight slot j & 1 and of
        // stage j + 2's fp32 patch into raw slot (j + 2) % 3 -- no registers, no compiler-inserted waits: the patch has two
        // stage times to arrive, the weights (L2-resident) one;  (2) split raw slot j % 3 into the three bf16 planes of plane
        // buffer j & 1;  (3) s_waitcnt vmcnt(patch instructions of this iteration): everything older -- stage j's weights,
        // stage j + 1's patch -- has landed; lgkmcnt(0); barrier.  (v2 staged through registers one stage ahead: the load
        // latency sat in every stage, profiles/r05_b3_v2_timeline.txt: a loading wave needed 4 000+ cycles per stage alone.)
        if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
        const int ltid = tid & 255;
        const __amdgpu_buffer_rsrc_t in_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t w_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16x8*>(a.w), (short)0, (int)a.w_bytes, 0x00020000);
        // patch DMA: instruction n of this wave covers pixels 16 (4 n + wave) .. + 15; lane = (pixel, quarter)
        int dpix[G::NRIW];
#pragma unroll
        for (int n = 0; n < G::NRIW; ++n) dpix[n] = 16 * (4 * n + wave) + (lane >> 2);
        unsigned voff[G::NRIW];
        auto place = [&](const B3Item& it, bool live) {
#pragma unroll
            for (int n = 0; n < G::NRIW; ++n) {
                const int r = dpix[n] / PW, c = dpix[n] - r * PW;
                const int y = it.y0 - 1 + r, x = it.x0 - 1 + c;
                const bool ok = live && dpix[n] < G::PIX && y >= 0 && y < a.H && x >= 0 && x < a.W;
                voff[n] = ok ? (unsigned)(((y * a.W + x) * CIN + (lane & 3) * 4) * 4) : 0x80000000u;
            }
        };
        auto dma_patch = [&](const B3Item& it, int s, int slot) {
            const int kd = s / NCH, ch = s - kd * NCH;
            const int dz = it.z + kd - KD / 2;
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)((((unsigned)(it.b * a.D + dz) * (unsigned)a.H * (unsigned)a.W) * CIN + ch * 16) * 4u));
            bf16x8* const dst0 = raw_ring + slot * G::RAWU;
#pragma unroll
            for (int n = 0; n < G::NRIW; ++n) {
                const unsigned off = voff[n];
                bf16x8* const dst = dst0 + (4 * n + wave) * 64;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (mvconv::lds_void*)dst, 16, off, soff, 0, 0);
            }
        };
        auto dma_weights = [&](const B3Item& it, int s, bool live, int slot) {
            if constexpr (!WREG) {
                const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)(((unsigned)s * a.nsplit + (unsigned)it.ns) * (unsigned)(WUNITS * 16)));
                bf16x8* const dst0 = w_ring + slot * WUNITS;
#pragma unroll
                for (int n = 0; n < NWI; ++n) {
                    const unsigned off = live ? (unsigned)(((4 * n + wave) * 64 + lane) * 16) : 0x80000000u;
                    bf16x8* const dst = dst0 + (4 * n + wave) * 64;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (mvconv::lds_void*)dst, 16, off, soff, 0, 0);
                }
            }
        };
        auto split = [&](int rslot, int pslot) {
            const bf16x8* const src = raw_ring + rslot * G::RAWU;
            unsigned char* const dstb = reinterpret_cast<unsigned char*>(planes + pslot * G::PATCH);
#pragma unroll
            for (int i = 0; i < G::NSU; ++i) {
                const int u = ltid + 256 * i;
                const int pix = u >> 2, q = u & 3;
                if (pix < G::PIX) {
                    const f32x4b v = __builtin_bit_cast(f32x4b, src[u]);
                    bf16x4 p1, p2, p3;
                    split4(v, p1, p2, p3);
                    unsigned char* const d = dstb + (((q >> 1) * G::HALF + pix) * 16 + (q & 1) * 8);
                    *reinterpret_cast<bf16x4*>(d) = p1;
                    *reinterpret_cast<bf16x4*>(d + PLANE * 16) = p2;
                    *reinterpret_cast<bf16x4*>(d + 2 * PLANE * 16) = p3;
                }
            }
        };
        struct Cursor { unsigned item; B3Item it; int s; bool live; };
        auto advance = [&](Cursor& c) {                        // (frozen, live = false, once the stream is exhausted)
            if (!c.live) return;
            if (c.s + 1 < c.it.s_end) {
                ++c.s;
            } else if (c.item + nwg < nitems) {
                c.item += nwg;
                c.it = decode(c.item);
                c.s = c.it.s_begin;
            } else {
                c.live = false;
            }
        };
        Cursor cw{first, decode(first), 0, true};              // the stage whose weights are requested / whose patch is split
        cw.s = cw.it.s_begin;
        Cursor cr = cw;                                        // the stage whose patch is requested: two ahead
        unsigned placed = 0xffffffffu;
        auto request_patch = [&](int slot) {
            if (cr.item != placed || !cr.live) {
                place(cr.it, cr.live);
                placed = cr.live ? cr.item : 0xffffffffu;
            }
            dma_patch(cr.it, cr.s, slot);
            advance(cr);
        };
        request_patch(0);                                      // stage 0
        request_patch(1);                                      // stage 1
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NRIW) : "memory");     // stage 0's patch has landed
        __builtin_amdgcn_s_barrier();                          // ... for every loading wave's quarter
        int k = 0;
        while (true) {
            B3_TL(0);
            dma_weights(cw.it, cw.s, cw.live, k & 1);          // stage k's weights
            request_patch((k + 2) % 3);                        // stage k + 2's patch
            B3_TL(1);
            split(k % 3, k & 1);
            B3_TL(2);
            advance(cw);
            const bool more = cw.live;
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G::NRIW) : "memory");
            __builtin_amdgcn_s_barrier();
            B3_TL(3);
            ++k;
            if (!more) break;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (no DMA may land after the workgroup's LDS is released)
        return;
    }

    // ---------------------------------------------------------------------------------------------------- compute waves
    if (a.prio == 1) __builtin_amdgcn_s_setprio(2);
    const int p = lane & 15, g = lane >> 4;
    // fragment addresses: lane (p, g) reads channels half g & 1 of tap 2 tp + (g >> 1) (tap 9 does not exist: its weights
    // are zero, the pixel operand re-reads tap 8 so that the product is 0 x (a value of this output's own footprint))
    int poff[5];
#pragma unroll
    for (int tp = 0; tp < 5; ++tp) {
        int tap = 2 * tp + (g >> 1);
        tap = tap > 8 ? 8 : tap;
        const int ky = tap / 3, kx = tap - 3 * ky;
        poff[tp] = (g & 1) * G::HALF + (ky + wave * TYQ) * PW + kx + p;
    }
    bf16x8 wreg[WREG ? 5 : 1][NTW][3];
    if constexpr (WREG) {
#pragma unroll
        for (int tp = 0; tp < 5; ++tp)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wreg[tp][j][pl] = a.w[((tp * NTW + j) * 3 + pl) * 64 + lane];
    }

    // Three accumulators per output tile, by magnitude of the terms: acc0 <- w1 p1; acc1 <- w1 p2, w2 p1 (2^-8); acc2 <- w2 p2,
    // w1 p3, w3 p1 (2^-16): the corrections are not rounded at the magnitude of the running sum, and no accumulator takes two
    // MFMAs in a row -- every step below visits all PT x NTW tiles before the next term, so dependent MFMAs are at least
    // PT x NTW issues apart (v1 chained five MFMAs into one accumulator, two chains interleaved: 2.3x the matrix time).
    f32x4b acc0[PT][NTW], acc1[PT][NTW], acc2[PT][NTW];
    __builtin_amdgcn_s_barrier();                              // (the loading waves' hand-over of stage 0's raw patch among themselves)
    int k = 0;
    for (unsigned item = first; item < nitems; item += nwg) {
        const B3Item it = decode(item);
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc0[i][j] = f32x4b{0.f, 0.f, 0.f, 0.f};
                acc1[i][j] = f32x4b{0.f, 0.f, 0.f, 0.f};
                acc2[i][j] = f32x4b{0.f, 0.f, 0.f, 0.f};
            }
        // (the epilogue's per-channel constants: requested here, so their latency passes under the MFMAs)
        f32x4b sc[NTW], sh[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int co = (it.ns * NTW + j) * 16 + 4 * g;
            sc[j] = *reinterpret_cast<const f32x4b*>(a.scale + co);
            sh[j] = *reinterpret_cast<const f32x4b*>(a.shift + co);
        }
        for (int s = it.s_begin; s < it.s_end; ++s, ++k) {
            __builtin_amdgcn_s_barrier();                      // stage k's planes and weights are in buffers k & 1
            B3_TL(0);
            const bf16x8* patch = planes + (k & 1) * G::PATCH;
            const bf16x8* wl = w_ring + (k & 1) * WUNITS;
            // Software pipeline over units = (tap pair, GP pixel tiles): 12 MFMAs each.  Unit u + 1's fragment reads are
            // issued BEFORE unit u's MFMAs and nothing crosses a unit boundary (sched_barrier): left to itself the scheduler
            // places a read one or two MFMAs ahead of its use (register pressure heuristics), i.e. ~30 cycles ahead of a
            // ~100-cycle LDS latency, and the matrix pipe ran at half rate (33 cycles per MFMA, profiles/r05_b3_v2_timeline.txt).
            bf16x8 wf[2][NTW][3], pf[2][GP][3];
            auto load_unit = [&](int u) {
                const int tp = u / UPT, iu = u - tp * UPT;
                if (iu == 0) {
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            if constexpr (WREG) wf[tp & 1][j][pl] = wreg[tp][j][pl];
                            else wf[tp & 1][j][pl] = wl[((tp * NTW + j) * 3 + pl) * 64 + lane];
                        }
                }
#pragma unroll
                for (int gi = 0; gi < GP; ++gi) {
                    const int i = iu * GP + gi;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) pf[u & 1][gi][pl] = patch[pl * PLANE + poff[tp] + (i >> 1) * PW + (i & 1) * 16];
                }
            };
            if (a.dbg & 2) continue;
            load_unit(0);
#pragma unroll
            for (int u = 0; u < NUNIT; ++u) {
                if (u + 1 < NUNIT) load_unit(u + 1);
                __builtin_amdgcn_sched_barrier(0);             // (the reads stay ahead of this unit's MFMAs)
                const int tp = u / UPT, iu = u - tp * UPT;
#define B3_STEP(ACC, WP, PP)                                                                                            \
    _Pragma("unroll") for (int gi = 0; gi < GP; ++gi) _Pragma("unroll") for (int j = 0; j < NTW; ++j)                    \
        ACC[iu * GP + gi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tp & 1][j][WP], pf[u & 1][gi][PP], ACC[iu * GP + gi][j], 0, 0, 0);
                B3_STEP(acc2, 2, 0)
                B3_STEP(acc1, 1, 0)
                B3_STEP(acc0, 0, 0)
                B3_STEP(acc2, 0, 2)
                B3_STEP(acc1, 0, 1)
                B3_STEP(acc2, 1, 1)
#undef B3_STEP
                __builtin_amdgcn_sched_barrier(0);
            }
            B3_TL(1);
        }
        // epilogue: lane = (pixel p of the 16-pixel tile, output channels 4 g .. 4 g + 3 of the N tile)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int co = (it.ns * NTW + j) * 16 + 4 * g;
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int y = it.y0 + wave * TYQ + (i >> 1), x = it.x0 + (i & 1) * 16 + p;
                if (y < a.H && x < a.W) {
                    const long o = ((((long)it.b * a.D + it.z) * a.H + y) * a.W + x) * a.cout + co;
                    f32x4b v = (acc0[i][j] + (acc1[i][j] + acc2[i][j])) * sc[j] + sh[j];
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    if (a.skip) v += *reinterpret_cast<const f32x4b*>(a.skip + o);
                    *reinterpret_cast<f32x4b*>(a.out + o) = v;
                }
            }
        }
#ifdef MVSTER_PROBES
        --k;
        B3_TL(2);
        ++k;
#endif
    }
}

template <int CIN, int NTW, int KD, int TYQ, bool WREG>
int launch_b3(const B3Args& a, int wpc, hipStream_t s) {
    using G = B3Geom<TYQ>;
    const size_t lds = (size_t)(3 * G::RAWU + 2 * G::PATCH + (WREG ? 0 : 2 * 1024 * NTW)) * 16;
    if (lds > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    static unsigned long big_done = 0;
    auto kern = conv_b3_kernel<CIN, NTW, KD, TYQ, WREG>;
    if (lds > 64 * 1024 && !mvconv::allow_big_lds(reinterpret_cast<const void*>(kern), big_done)) return MVSTER_ERR_LAUNCH;
    const int cus = mvconv::num_cus();
    if (cus <= 0) return MVSTER_ERR_LAUNCH;
    const unsigned fit = (unsigned)((160 * 1024) / lds);                     // workgroups one CU's LDS holds
    unsigned per_cu = wpc > 0 ? (unsigned)wpc : 1u;
    per_cu = per_cu > fit ? fit : per_cu;
    per_cu = per_cu > 2 ? 2 : per_cu;                                        // (512 threads x 2 = the CU's 16 wave slots at <= 128 VGPRs)
    const unsigned nitems = a.ntiles * a.nsplit;
    const unsigned grid = nitems < (unsigned)cus * per_cu ? nitems : (unsigned)cus * per_cu;
    MV_NOTE_KERNEL("conv_b3_kernel<%d, %d, %d, %d, %s>", CIN, NTW, KD, TYQ, WREG ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    return mv_check_launch();
}

template <int CIN, int NTW>
int launch_b3_shape(const B3Args& a, int kd, int tyq, int wpc, hipStream_t s) {
    // (8-row tiles fit the LDS only where the weights live in registers: raw ring 72 KB + planes 66 KB)
    if (kd == 1) {
        if constexpr (CIN == 16 && NTW == 1)                  // one stage per item: the weights stay in registers
            return tyq == 1 ? launch_b3<CIN, NTW, 1, 1, true>(a, wpc, s) : launch_b3<CIN, NTW, 1, 2, true>(a, wpc, s);
        else
            return launch_b3<CIN, NTW, 1, 1, false>(a, wpc, s);
    }
    return launch_b3<CIN, NTW, 3, 1, false>(a, wpc, s);
}

}  // namespace
#endif  // MVSTER_PROBES

namespace mvconv {

#ifndef MVSTER_PROBES
int dispatch_b3(const ConvArgs&, int, int, hipStream_t) { return MVSTER_ERR_UNSUPPORTED; }   // (probe library only)
#else
// variant 11 of mvster_conv_mfma; `wpk` = the pre-split bf16 fragments of conv_plan.py:pack_b3; mt = rows per compute wave
// (1 | 2); wpc = workgroups per CU (0 = 1)
int dispatch_b3(const ConvArgs& c, int mt, int wpc, hipStream_t s) {
    if (c.nclass != 1 || c.sd != 1 || c.sh != 1 || c.sw != 1 || c.osd != 1 || c.osh != 1 || c.osw != 1) return MVSTER_ERR_UNSUPPORTED;
    const int kd = c.kd[0];
    if (c.kh[0] != 3 || c.kw[0] != 3 || (kd != 1 && kd != 3) || c.ph[0] != 1 || c.pw[0] != 1 || c.pd[0] != kd / 2)
        return MVSTER_ERR_UNSUPPORTED;
    if (c.Do != c.Di || c.Ho != c.Hi || c.Wo != c.Wi || c.prob_w || c.skip_mode == 2) return MVSTER_ERR_UNSUPPORTED;
    if (c.cout != 16 && c.cout != 32 && c.cout != 64) return MVSTER_ERR_UNSUPPORTED;
    if ((long)c.B * c.Do * c.Ho * c.Wo * c.cout >= (1L << 31)) return MVSTER_ERR_SHAPE;
    const int ntw = c.cout == 16 ? 1 : 2;
    const int tyq = (mt == 2 && c.cin == 16 && ntw == 1 && kd == 1) ? 2 : 1;
    B3Args a;
    a.in = c.in; a.w = reinterpret_cast<const bf16x8*>(c.wpk); a.scale = c.scale; a.shift = c.shift;
    a.skip = c.skip_mode == 1 ? c.skip : nullptr; a.out = c.out;
    a.B = c.B; a.D = c.Do; a.H = c.Ho; a.W = c.Wo; a.cout = c.cout; a.relu = c.relu;
    a.in_bytes = c.in_bytes;
    a.w_bytes = (unsigned)((size_t)kd * (c.cin / 16) * (c.cout / (16 * ntw)) * 1024 * ntw * 16);
    a.prio = (wpc >> 2) & 3;
    a.dbg = (wpc >> 4) & 3;
    wpc &= 3;
    a.tiles_x = (unsigned)((c.Wo + 31) / 32);
    a.tiles_y = (unsigned)((c.Ho + 4 * tyq - 1) / (4 * tyq));
    a.nsplit = (unsigned)(c.cout / (16 * ntw));
    const long ntiles = (long)a.tiles_x * a.tiles_y * c.Do * c.B;
    if (ntiles * a.nsplit >= (1L << 31)) return MVSTER_ERR_SHAPE;
    a.ntiles = (unsigned)ntiles;
    switch (c.cin) {
        case 16: return ntw == 1 ? launch_b3_shape<16, 1>(a, kd, tyq, wpc, s) : launch_b3_shape<16, 2>(a, kd, tyq, wpc, s);
        case 32: return ntw == 1 ? launch_b3_shape<32, 1>(a, kd, tyq, wpc, s) : launch_b3_shape<32, 2>(a, kd, tyq, wpc, s);
        case 64: return ntw == 1 ? launch_b3_shape<64, 1>(a, kd, tyq, wpc, s) : launch_b3_shape<64, 2>(a, kd, tyq, wpc, s);
        default: return MVSTER_ERR_UNSUPPORTED;
    }
}

#endif  // MVSTER_PROBES

}  // namespace mvconv

#ifdef MVSTER_PROBES
extern "C" int mvster_b3_timeline(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_b3tl), &buf, sizeof(buf)) == hipSuccess ? MVSTER_OK : MVSTER_ERR_LAUNCH;
}
#endif
