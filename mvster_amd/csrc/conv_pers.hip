// Persistent, LDS-DMA double-buffered implicit-GEMM convolution (variant 5 of mvster_conv_mfma).
//
// Why: s_memtime stamps of conv_lds_kernel (profiles/r03_a_conv_lds_timeline.txt) show a wavefront of a one-chunk
// 16 -> 16 layer alive for ~19 500 cycles of which the 72 MFMAs are 29 % (5 700 cycles against 2 304 of pipe time):
// 32 % goes into the prologue (kernel arguments, tile decode, staging addresses, weight / BatchNorm fetches -- paid per
// 128-pixel tile), 19 % into waiting for the staged patch and the barrier, 20 % into the epilogue.  This kernel keeps a
// workgroup alive over many tiles:
//   * weights, scale/shift and all tile-independent index arithmetic are set up ONCE per workgroup; the weights stay
//     in LDS (or, for the 16 -> 16 3x3 layers, in 36 registers) for its lifetime;
//   * the input patch of tile t+1 is fetched by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass,
//     zero padding = the descriptor's range check) into the second LDS buffer while tile t's MFMAs run;
//   * one barrier per tile; the epilogue is a specialised float4 path (scale/shift, ReLU, optional same-shape skip).
// Same packed weights and fused epilogue semantics as the other variants; K order = tap-major, channel-minor (the packed
// order), i.e. bit-identical to the direct kernel.
//
// Workgroup = 4 waves = TY x 32 output pixels of one (b, z) slice, TY = 2*MT; wave w owns M tiles w*MT .. w*MT+MT-1
// (tile t = row t >> 1, columns (t & 1)*16 ..).  LDS patch layout per 16-channel chunk: two planes (plane = quad >> 1) of
// [pixel][2 quads] float4 -- the conflict-free layout of conv_lds_kernel; for stride-2 layers the pixels of a row are
// split into even / odd columns so that a tap again reads 16 consecutive pixels.  LDS-DMA writes are lane-linear
// (64 consecutive float4 per wave instruction), so the layout is produced by the SOURCE address decode.
//
// Reference layers: Conv2d / ConvBnReLU3D of models/mvs4net_utils.py:116-123, :224-251 as used by FPN4 (:419-502) and
// reg2d / reg3d (:870-965).
#include "conv_args.hpp"

namespace mvconv {
namespace {


// Probe build only (make timeline; scripts/conv_pers_timeline.py): per wavefront and tile, s_memtime at the phase
// boundaries.  Record = 8 x u64 per (workgroup, wave, tile slot < 16): t0 loop top, t1 DMA of the next tile issued,
// t2 MFMAs issued, t3 past the barrier (next patch landed), t4 epilogue stores issued, [5] HW_ID, [6] XCC_ID, [7] tile.
#ifdef MVSTER_TIMELINE
__device__ unsigned long long* g_ptl = nullptr;
#define MV_PTL(k)                                                                                                       \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (g_ptl && lane == 0 && it < 16)                                                                              \
            g_ptl[(((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 16 + it) * 8 + (k)] =                    \
                (k) < 5 ? __builtin_amdgcn_s_memtime() : (k) == 5 ? __builtin_amdgcn_s_getreg(63492)                    \
                                                         : (k) == 6 ? __builtin_amdgcn_s_getreg(63508) : tile;          \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
// conv_pp_kernel: per wavefront and half-phase (< 16): [0] top of the half-phase, [1] work done (before the barrier),
// [2] 1 = MFMA half-phase with a tile, 2 = prepare half-phase, 0 = idle, [3] HW_ID.
#define MV_PPTL(k, v)                                                                                                   \
    do {                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if (g_ptl && lane == 0 && hp < 16)                                                                              \
            g_ptl[(((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave8) * 16 + hp) * 4 + (k)] = (v);           \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    } while (0)
#else
#define MV_PTL(k)
#define MV_PPTL(k, v)
#endif

// MT, NT: register tile (M tiles x N tiles of 16) per wave; KW = KH in {3, 5}; SW = SH in {1, 2}; NCH = cin / 16;
// KD in {1, 3}; WREG: the layer's weights live in registers (KD*KW*KW*NCH*NT float4 per lane), else in LDS;
// PF = prefetch distance of the tap pipeline (taps); SKIP: a same-shape tensor is added in the epilogue.
// LD: waves 4-7 of a 512-thread workgroup issue the LDS-DMA (a loop of their own, same barriers), waves 0-3 compute -- a wave
// that requests data at the rate HBM delivers it stalls ~300 cycles per kilobyte at issue (conv_wgrad_pers.hip measured it).
template <int MT, int NT, int KW, int SW, int NCH, int KD, bool WREG, int PF, bool SKIP, bool LD>
__global__ void __launch_bounds__(LD ? 512 : 256) conv_pers_kernel(ConvArgs a, PersArgs p) {
    using G = PersGeom<MT, KW, SW, KD>;
    constexpr int TY = G::TY, KH = G::KH, PW = G::PW, PH = G::PH, PWH = G::PWH, PLANE = G::PLANE, NBLK = G::NBLK;
    constexpr int NTAP = KD * KH * KW, TAPS2D = KH * KW;
    constexpr int BUF = NCH * 2 * PLANE;                    // float4 per patch buffer
    constexpr int NI = NCH * 2 * NBLK;                      // DMA wave-instructions per tile
    constexpr int NIW = (NI + 3) / 4;
    constexpr int CIN = NCH * 16;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const scratch = lds + 2 * BUF;                  // 64 float4: target of the surplus DMA slots (NI % 4 != 0)
    f32x4v* const wl = scratch + 64;                        // [tap][chunk][nt][lane]  (unused with WREG)

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;
    const bool loader = LD && wave8 >= 4, loads = !LD || loader;
    const int lm = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.y * NT;
    // Two workgroups sharing a CU start together and, left alone, stay IN phase: both in their MFMA phase (sharing the
    // pipe), then both in their address / wait / store phase (pipe idle) -- the arbitration is symmetric.  A static
    // priority for the waves in odd slots of their SIMD breaks the tie: the favoured workgroup owns the pipe during its
    // MFMA phase, the other one runs its own in the gaps, i.e. they settle in anti-phase.
    if (p.prio) {
        const unsigned slot = __builtin_amdgcn_s_getreg(63492) & 15u;      // HW_ID.wave_id
        if ((slot & 3u) == 1u) __builtin_amdgcn_s_setprio(1);
        if ((slot & 3u) == 2u) __builtin_amdgcn_s_setprio(2);
        if ((slot & 3u) == 3u) __builtin_amdgcn_s_setprio(3);
    }
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);

    // ---- tile-independent per-lane part of the DMA addresses ------------------------------------------------------
    // instruction i = wave + 4n -> (chunk c, plane pl, block blk); lane -> slot blk*64 + lane of that plane
    // slot -> (patch row, px, quad parity).  dbase = byte offset of the lane's 16 bytes relative to the patch origin
    // (0x80000000 for slots that hold no pixel: stays out of range after adding any tile origin < 2^31);
    // dpos = px | py << 8 | pz << 16 for the border tiles, whose out-of-image lanes are sent out of range one by one.
    unsigned dbase[NIW];
    int dpos[NIW];
#pragma unroll
    for (int n = 0; n < NIW; ++n) {
        const int i = wave + 4 * n;
        const int c = i / (2 * NBLK), r = i - c * 2 * NBLK, pl = r / NBLK, blk = r - pl * NBLK;
        const int s = blk * 64 + lane;
        const int q1 = s & 1;
        int prow, px;
        if (SW == 1) {
            const int pix = s >> 1;
            prow = pix / PW;
            px = pix - prow * PW;
        } else {
            int t = s >> 1;
            const int xh = t % PWH;
            t /= PWH;
            prow = t >> 1;
            px = 2 * xh + (t & 1);
        }
        const bool valid = i < NI && prow < G::ROWS && px < PW;
        const int pz = KD == 1 ? 0 : prow / PH, py = prow - pz * PH;
        dpos[n] = px | (py << 8) | (pz << 16);
        dbase[n] = valid ? (unsigned)(((pz * a.Hi + py) * a.Wi + px) * (CIN * 4) + (c * 16 + pl * 8 + q1 * 4) * 4) : 0x80000000u;
    }

    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        // (branch-free: a select on the divisor being 1 would split the loop body into several basic blocks)
        auto div = [&](unsigned n, int k) -> unsigned { return ((__umulhi(n, p.mul[k]) >> p.shr[k]) & ~p.one[k]) | (n & p.one[k]); };
        unsigned q = div(tile, 0);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = div(q, 1);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = div(q2, 2);
        t.zo = (int)(q2 - q3 * (unsigned)a.Do);
        t.b = (int)q3;
        return t;
    };

    // (branch-free: on interior tiles every lane's bounds test passes; `live` = false sends the whole patch out of range --
    //  used for the DMA slot of a tile that does not exist, so that the loop body stays one basic block)
    auto dma_tile = [&](const TilePos& t, int buf, bool live) {
        const int iz0 = t.zo * a.sd - a.pd[0], iy0 = t.ty0 * SW - a.ph[0], ix0 = t.tx0 * SW - a.pw[0];
        // byte offset of the patch origin (may be negative on border tiles: 32-bit wrap-around arithmetic)
        const unsigned origin = (unsigned)((((t.b * a.Di + iz0) * a.Hi + iy0) * a.Wi + ix0) * (CIN * 4));
        const unsigned wi = live ? (unsigned)a.Wi : 0u;
        f32x4v* const dst0 = lds + buf * BUF;
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int i = wave + 4 * n;
            {
                const int ix = ix0 + (dpos[n] & 255), iy = iy0 + ((dpos[n] >> 8) & 255), iz = iz0 + (dpos[n] >> 16);
                bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < wi;
                if (KD > 1) ok = ok && (unsigned)iz < (unsigned)a.Di;
                // (named operands: hipcc 7.2 silently drops the kernel's host stub when this builtin is handed an arithmetic
                //  expression as its offset)
                const unsigned off = ok ? dbase[n] + origin : 0x80000000u;
                // (a wave without an n-th slot still issues it -- into the scratch block, every lane out of range: no
                //  wave-dependent branch in the loop body)
                f32x4v* const dst = (NI % 4 == 0 || n + 1 < NIW || i < NI) ? dst0 + (i / NBLK) * PLANE + (i % NBLK) * 64 : scratch;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
            }
        }
    };

    // ---- once per workgroup: weights, scale / shift, A-read bases ------------------------------------------------
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);          // this workgroup's tiles: tile, tile + nwg, ...  (see host side)
    TilePos pos = decode_tile(tile < p.ntiles ? tile : 0);
    if (tile < p.ntiles && loads) dma_tile(pos, 0, true);
    f32x4v wreg[WREG ? NTAP * NCH * NT : 1];
    const long wstep = (long)a.ntile_total * 256;          // floats per K step of the packed weights
    if (WREG) {
#pragma unroll
        for (int s = 0; s < NTAP * NCH; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wreg[s * NT + nt] = *reinterpret_cast<const f32x4v*>(a.wpk + s * wstep + ((long)(nt0 + nt) * 64 + lane) * 4);
    } else {
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wpk), (short)0, (int)(NTAP * NCH * wstep * 4), 0x00020000);
        for (int i = loads ? wave : NTAP * NCH * NT; i < NTAP * NCH * NT; i += 4) {
            const int s = i / NT, nt = i - s * NT;
            const unsigned off = (unsigned)((s * wstep + (long)(nt0 + nt) * 256) * 4) + lane * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wl + i * 64), 16, off, 0, 0, 0);
        }
    }
    f32x4v scv[NT], shv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = (nt0 + nt) * 16 + lq * 4;
        scv[nt] = *reinterpret_cast<const f32x4v*>(a.scale + n0);
        shv[nt] = *reinterpret_cast<const f32x4v*>(a.shift + n0);
    }
    // float4 index (inside a chunk's plane pair) of this lane's A operand for tap (0,0,0) of each of its M tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, x = (t & 1) * 16 + lm;
        const int rs = G::ROWSLOTS;
        abase[mt] = (SW == 1 ? row * rs + x * 2 : row * 2 * rs + x * 2) + (lq >> 1) * PLANE + (lq & 1);
    }
    // output (and skip) byte offset of this lane's 4 channels of each M tile, relative to the tile's first pixel
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)p.out_bytes : 0, 0x00020000);
    unsigned obase[MT];
    int orc[MT];            // row | col << 8 of the M tile's pixel inside the workgroup tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, col = (t & 1) * 16 + lm;
        orc[mt] = row | (col << 8);
        obase[mt] = (unsigned)((row * a.Wo + col) * a.cout + nt0 * 16 + lq * 4) * 4u;
    }
    auto tap_off = [&](int kz, int ky, int kx) -> int {     // float4 offset of a tap relative to abase (compile-time)
        const int prow = kz * PH + ky;
        return SW == 1 ? prow * G::ROWSLOTS + kx * 2 : prow * G::ROWSLOTS + (kx & 1) * PWH * 2 + (kx >> 1) * 2;
    };

    __syncthreads();        // (waits vmcnt(0): first patch and the weights have landed)
    if (loader) {
        // the loading waves: request tile t+1 while the compute waves work on tile t, wait until it has landed, meet them
        for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
            const bool has_next = tile + nwg < p.ntiles;
            pos = decode_tile(has_next ? tile + nwg : tile);
            dma_tile(pos, (it & 1) ^ 1, has_next);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // The loop body is ONE basic block (2-D kernels): the DMA of tile t+1, the MFMAs of tile t and the epilogue of tile
    // t-1 are independent instruction streams that the scheduler interleaves -- at one wavefront per SIMD nothing else
    // could fill the matrix pipe's shadow.  Stores stay below the DMA issues (sched_barrier) so that a counted
    // vmcnt(stores) before the barrier means "the next patch has landed" without draining the stores.
    f32x4v pacc[MT][NT], pskv[MT][NT];
    unsigned pooff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        pooff[mt] = 0x80000000u;                 // nothing to store in the first round
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pacc[mt][nt] = pskv[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    }
    auto epilogue = [&](const f32x4v (&accv)[MT][NT], const f32x4v (&skvv)[MT][NT], const unsigned (&off)[MT]) {
        // the accumulator is D^T (weights in the A slot): 4 consecutive output channels of one voxel per lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4v v = accv[mt][nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = fmaf(v[j], scv[nt][j], shv[nt][j]);
                    if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                    if (SKIP) v[j] += skvv[mt][nt][j];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, off[mt] + nt * 64, 0, MV_STORE_AUX);
            }
        }
    };

    for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
        const int cur = it & 1;
        MV_PTL(0);
        MV_PTL(5);
        MV_PTL(6);
        MV_PTL(7);
        const TilePos here = pos;
        // this lane's output offsets (out of range = dropped by the hardware) and the skip values, fetched ahead
        const unsigned oorigin = (unsigned)((((here.b * a.Do + here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
        unsigned ooff[MT];
        f32x4v skv[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool ok = here.ty0 + (orc[mt] & 255) < a.Ho && here.tx0 + (orc[mt] >> 8) < a.Wo;
            ooff[mt] = ok ? obase[mt] + oorigin : 0x80000000u;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                skv[mt][nt] = SKIP ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooff[mt] + nt * 64, 0, 0))
                                   : (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
        const bool has_next = tile + nwg < p.ntiles;
        pos = decode_tile(has_next ? tile + nwg : tile);
        if (!LD) dma_tile(pos, cur ^ 1, has_next);
        MV_PTL(1);
        // VMEM may not cross: the DMA issues stay above, the stores of the deferred epilogue below
        __builtin_amdgcn_sched_barrier(0x78F);
        if (KD == 1) epilogue(pacc, pskv, pooff);

        f32x4v acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};

        const f32x4v* patch = lds + cur * BUF;
        // depth taps that fall entirely into the zero padding of this output slice are skipped
        const int kz_lo = KD == 1 ? 0 : max(0, a.pd[0] - here.zo * a.sd), kz_hi = KD == 1 ? 1 : min(KD, a.Di + a.pd[0] - here.zo * a.sd);
#pragma unroll
        for (int kz = 0; kz < KD; ++kz) {
            if (kz < kz_lo || kz >= kz_hi) continue;
            // software pipeline over the KH*KW taps of this depth slice: PF taps of operands in flight
            f32x4v A[PF + 1][NCH][MT], Bv[PF + 1][NCH][NT];
            auto load_tap = [&](int t2, f32x4v (&Aa)[NCH][MT], f32x4v (&Bb)[NCH][NT]) {
                const int ky = t2 / KW, kx = t2 - ky * KW;
                const int to = tap_off(kz, ky, kx);
                const int tap = kz * TAPS2D + t2;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) Aa[c][mt] = patch[abase[mt] + c * 2 * PLANE + to];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        Bb[c][nt] = WREG ? wreg[(tap * NCH + c) * NT + nt] : wl[((tap * NCH + c) * NT + nt) * 64 + lane];
                }
            };
#pragma unroll
            for (int t2 = 0; t2 < PF && t2 < TAPS2D; ++t2) load_tap(t2, A[t2 % (PF + 1)], Bv[t2 % (PF + 1)]);
#pragma unroll
            for (int t2 = 0; t2 < TAPS2D; ++t2) {
                if (t2 + PF < TAPS2D) load_tap(t2 + PF, A[(t2 + PF) % (PF + 1)], Bv[(t2 + PF) % (PF + 1)]);
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bv[t2 % (PF + 1)][c][nt][j], A[t2 % (PF + 1)][c][mt][j],
                                                                                   acc[mt][nt], 0, 0, 0);
            }
            if (KD > 1 && kz == 0) epilogue(pacc, pskv, pooff);     // 3-D kernels: behind the first depth slice's MFMAs
        }
        if (KD > 1 && kz_lo > 0) epilogue(pacc, pskv, pooff);       // (that slice was padding: not stored yet)
        MV_PTL(2);
        // the next tile's patch has landed once at most the stores issued after it are outstanding; then everyone is
        // done reading this tile's patch
        if (!LD) {
            if (KD == 1) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * NT) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): (the barrier builtin alone does not wait for the LDS reads)
        __builtin_amdgcn_s_barrier();
        MV_PTL(3);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            pooff[mt] = ooff[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                pacc[mt][nt] = acc[mt][nt];
                pskv[mt][nt] = skv[mt][nt];
            }
        }
        MV_PTL(4);
    }
    epilogue(pacc, pskv, pooff);
}

// ------------------------------------------------------------------------------------------------------------------
// 8 -> 16 channels, stride 2 (FPN conv1.0: 5x5 on the full-resolution maps; reg2d conv1: 3x3): the persistent frame for
// EIGHT input channels.  A 16-wide K step of the packed weights is two kernel taps x 8 channels, so lane (lm, lq) reads
// tap 2s + (lq >> 1), channel quad lq & 1: one plane of [pixel][2 quads] per patch, the tap pair's two LDS offsets picked per
// lane once.  All 13 (5) weight fragments in registers; four compute waves (2 M tiles each: 4 x 32 output pixels per tile),
// waves 4-7 issue the LDS-DMA (the input is the largest tensor of the forward and comes from HBM).  The direct kernel these
// layers ran on feeds every K step from L1 (2 KB per 4 MFMAs and wave = the TCP's 64 B/clk): 0.30 of the MFMA peak.
// Same K order and epilogue arithmetic as the direct kernel: bit-identical.
// ------------------------------------------------------------------------------------------------------------------
template <int KW, bool SKIP>
__global__ void __launch_bounds__(512) conv_pers8_kernel(ConvArgs a, PersArgs p) {
    constexpr int MT = 2, SW = 2;
    using G = PersGeom<MT, KW, SW, 1>;
    constexpr int TY = G::TY, PW = G::PW, PWH = G::PWH, PLANE = G::PLANE, NBLK = G::NBLK;
    constexpr int NTAP = KW * KW, NKS = (NTAP * 8 + 15) / 16;             // K steps of 16 = tap pairs
    constexpr int NIW = (NBLK + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);              // two patches of one plane each
    f32x4v* const scratch = lds + 2 * PLANE;

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const int lm = lane & 15, lq = lane >> 4;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);

    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        auto div = [&](unsigned n, int k) -> unsigned { return ((__umulhi(n, p.mul[k]) >> p.shr[k]) & ~p.one[k]) | (n & p.one[k]); };
        unsigned q = div(tile, 0);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = div(q, 1);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = div(q2, 2);
        t.zo = (int)(q2 - q3 * (unsigned)a.Do);
        t.b = (int)q3;
        return t;
    };
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);

    if (loader) {
        // instruction i = wave + 4n -> block of 64 slots; slot -> (row, column parity, column pair index, quad)
        unsigned dbase[NIW];
        int dpos[NIW];
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int i = wave + 4 * n;
            const int s = i * 64 + lane;
            const int q1 = s & 1;
            int t = s >> 1;
            const int xh = t % PWH;
            t /= PWH;
            const int prow = t >> 1, px = 2 * xh + (t & 1);
            const bool valid = i < NBLK && prow < G::ROWS && px < PW;
            dpos[n] = px | (prow << 8);
            dbase[n] = valid ? (unsigned)((prow * a.Wi + px) * 32 + q1 * 16) : 0x80000000u;
        }
        auto dma_tile = [&](const TilePos& t, int buf, bool live) {
            const int iy0 = t.ty0 * SW - a.ph[0], ix0 = t.tx0 * SW - a.pw[0];
            const unsigned origin = (unsigned)((((t.b * a.Di + t.zo) * a.Hi + iy0) * a.Wi + ix0) * 32);
            const unsigned wi = live ? (unsigned)a.Wi : 0u;
            f32x4v* const dst0 = lds + buf * PLANE;
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const int i = wave + 4 * n;
                const int ix = ix0 + (dpos[n] & 255), iy = iy0 + (dpos[n] >> 8);
                const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < wi;
                const unsigned off = ok ? dbase[n] + origin : 0x80000000u;
                f32x4v* const dst = (NBLK % 4 == 0 || n + 1 < NIW || i < NBLK) ? dst0 + i * 64 : scratch;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
            }
        };
        if (tile < p.ntiles) dma_tile(decode_tile(tile), 0, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
            const bool has_next = tile + nwg < p.ntiles;
            dma_tile(decode_tile(has_next ? tile + nwg : tile), (it & 1) ^ 1, has_next);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---- compute waves ---------------------------------------------------------------------------------------------------
    f32x4v wreg[NKS];
#pragma unroll
    for (int s2 = 0; s2 < NKS; ++s2) wreg[s2] = *reinterpret_cast<const f32x4v*>(a.wpk + (long)s2 * a.ntile_total * 256 + lane * 4);
    const f32x4v scv = *reinterpret_cast<const f32x4v*>(a.scale + lq * 4), shv = *reinterpret_cast<const f32x4v*>(a.shift + lq * 4);
    // float4 offset of this lane's tap of every tap pair, relative to its pixel's slot for tap (0, 0)
    int toff[NKS];
#pragma unroll
    for (int s2 = 0; s2 < NKS; ++s2) {
        int tap = 2 * s2 + (lq >> 1);
        tap = tap < NTAP ? tap : NTAP - 1;                  // (the padded half of the last K step: zero weights, any patch slot)
        const int ky = tap / KW, kx = tap - ky * KW;
        toff[s2] = ky * G::ROWSLOTS + (kx & 1) * PWH * 2 + (kx >> 1) * 2 + (lq & 1);
    }
    int abase[MT];
    unsigned obase[MT];
    int orc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, col = (t & 1) * 16 + lm;
        abase[mt] = row * 2 * G::ROWSLOTS + col * 2;
        orc[mt] = row | (col << 8);
        obase[mt] = (unsigned)((row * a.Wo + col) * a.cout + lq * 4) * 4u;
    }
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)p.out_bytes : 0, 0x00020000);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // (the loaders' first patch has landed)
    for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
        const TilePos here = decode_tile(tile);
        const unsigned oorigin = (unsigned)((((here.b * a.Do + here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
        unsigned ooff[MT];
        f32x4v skv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool ok = here.ty0 + (orc[mt] & 255) < a.Ho && here.tx0 + (orc[mt] >> 8) < a.Wo;
            ooff[mt] = ok ? obase[mt] + oorigin : 0x80000000u;
            skv[mt] = SKIP ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooff[mt], 0, 0)) : (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
        const f32x4v* patch = lds + (it & 1) * PLANE;
        f32x4v acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        // K order = the packed order (tap-major, channel-minor), operands one K step ahead
        f32x4v A[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) A[0][mt] = patch[abase[mt] + toff[0]];
#pragma unroll
        for (int s2 = 0; s2 < NKS; ++s2) {
            if (s2 + 1 < NKS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) A[(s2 + 1) & 1][mt] = patch[abase[mt] + toff[s2 + 1]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s2][j], A[s2 & 1][mt][j], acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4v v = acc[mt];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = fmaf(v[j], scv[j], shv[j]);
                if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                if (SKIP) v[j] += skv[mt][j];
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, ooff[mt], 0, MV_STORE_AUX);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // done reading this patch
        __builtin_amdgcn_s_barrier();
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Transposed 1x3x3 stride (1,2,2) convolutions (reg2d's conv7 / conv9: 64 -> 32, 32 -> 16, with the U-Net skip added at the
// output resolution; since round 6 also 16 -> 8, the last transposed layer and the stride-2 layers' input gradient in training).  The direct kernel runs them as four output-parity classes, each a 1-, 2-, 2- or 4-tap stride-1
// convolution over the input lattice in its own set of workgroups: 2 to 8 K steps per workgroup tile -- all prologue, no
// steady state (16-25 us at stage 4 against 4-8 us of HBM / MFMA time).  Here a persistent workgroup stages a 4 x 32 input
// tile (+ one halo row and column: the taps reach i and i + 1) once and computes ALL FOUR classes from it, 18 K steps per
// chunk set; the classes' packed weights (their own K order: bit-identical to the direct kernel) stay in LDS; four compute
// waves (2 M tiles each), four loading waves.
// ------------------------------------------------------------------------------------------------------------------
// KS = 5 (round 6): the 1x5x5 stride (1,2,2) form -- the input gradients of the FPN's 5x5 stride-2 convolutions, which ran on the
// direct kernel (16 -> 8 at [10, 256, 320]: 217 us for 65 us of HBM time).  The output parities take 3 / 2 taps per axis (classes of
// 9 / 6 / 6 / 4 taps, the even ones padded by one): the tile is staged with a one-pixel ring, 6 x 34 input pixels, and an odd
// class starts one row / column further in.
template <int NCH, bool SKIP, int KS>
__global__ void __launch_bounds__(512) conv_tpers_kernel(ConvArgs a, PersArgs p) {
    static_assert(KS == 3 || KS == 5, "1x3x3 or 1x5x5, stride (1,2,2)");
    constexpr int MT = 2;                                   // (one N tile per workgroup: blockIdx.y)
    constexpr bool K5 = KS == 5;
    constexpr int KT = K5 ? 3 : 2;                          // most taps per axis of a class
    constexpr int PADO = K5 ? 1 : 0;                        // the staged tile starts this far in front of the output tile's input pixel
    using G = PersGeom<MT, KT, 1, 1>;                       // 5 x 33 (6 x 34) input pixels per tile
    constexpr int TY = G::TY, PW = G::PW, PLANE = G::PLANE, NBLK = G::NBLK, RS = G::ROWSLOTS;
    constexpr int BUF = NCH * 2 * PLANE;
    constexpr int NI = NCH * 2 * NBLK, NIW = (NI + 3) / 4;
    constexpr int CIN = NCH * 16;
    // (K steps of all four classes: (1 + 2 + 2 + 4) or (9 + 6 + 6 + 4) taps x NCH chunks = 9 (25) * NCH kilobytes of weights per N tile)
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw);
    f32x4v* const scratch = lds + 2 * BUF;
    f32x4v* const wl = scratch + 64;                        // [class steps][lane]

    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave8 & 3;
    const bool loader = wave8 >= 4;
    const int lm = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.y;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        auto div = [&](unsigned n, int k) -> unsigned { return ((__umulhi(n, p.mul[k]) >> p.shr[k]) & ~p.one[k]) | (n & p.one[k]); };
        unsigned q = div(tile, 0);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = div(q, 1);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = div(q2, 2);
        t.zo = (int)(q2 - q3 * (unsigned)a.Do);
        t.b = (int)q3;
        return t;
    };
    const unsigned nwg = gridDim.x;
    unsigned tile = xcd_remap(blockIdx.x, nwg);

    if (loader) {
        unsigned dbase[NIW];
        int dpos[NIW];
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int i = wave + 4 * n;
            const int c = i / (2 * NBLK), r = i - c * 2 * NBLK, pl = r / NBLK, blk = r - pl * NBLK;
            const int s = blk * 64 + lane;
            const int q1 = s & 1, pix = s >> 1;
            const int prow = pix / PW, px = pix - prow * PW;
            const bool valid = i < NI && prow < G::ROWS;
            dpos[n] = px | (prow << 8);
            dbase[n] = valid ? (unsigned)((prow * a.Wi + px) * (CIN * 4) + (c * 16 + pl * 8 + q1 * 4) * 4) : 0x80000000u;
        }
        auto dma_tile = [&](const TilePos& t, int buf, bool live) {
            // (the ring of the 5x5 form may put the origin in front of the tensor: 32-bit wrap-around, only in-image lanes use it)
            const unsigned origin = (unsigned)((((t.b * a.Di + t.zo) * a.Hi + t.ty0 - PADO) * a.Wi + t.tx0 - PADO) * (CIN * 4));
            const unsigned wi = live ? (unsigned)a.Wi : 0u;
            f32x4v* const dst0 = lds + buf * BUF;
#pragma unroll
            for (int n = 0; n < NIW; ++n) {
                const int i = wave + 4 * n;
                const int ix = t.tx0 - PADO + (dpos[n] & 255), iy = t.ty0 - PADO + (dpos[n] >> 8);
                const bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < wi;
                const unsigned off = ok ? dbase[n] + origin : 0x80000000u;
                f32x4v* const dst = (NI % 4 == 0 || n + 1 < NIW || i < NI) ? dst0 + (i / NBLK) * PLANE + (i % NBLK) * 64 : scratch;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
            }
        };
        // the four classes' packed weights: class c starts at woff[c], nsteps[c] K steps of ntile_total x 256 floats
        {
            const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.wpk), (short)0, (int)((a.woff[3] + (long)a.nsteps[3] * a.ntile_total * 256) * 4), 0x00020000);
            int base = 0;
            for (int c = 0; c < 4; ++c) {
                for (int sidx = wave; sidx < a.nsteps[c]; sidx += 4) {
                    const unsigned off = (unsigned)((a.woff[c] + ((long)sidx * a.ntile_total + nt0) * 256) * 4) + lane * 16;
                    f32x4v* const dst = wl + (base + sidx) * 64;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
                }
                base += a.nsteps[c];
            }
        }
        if (tile < p.ntiles) dma_tile(decode_tile(tile), 0, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
            const bool has_next = tile + nwg < p.ntiles;
            dma_tile(decode_tile(has_next ? tile + nwg : tile), (it & 1) ^ 1, has_next);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---- compute waves ---------------------------------------------------------------------------------------------------
    const f32x4v scv = *reinterpret_cast<const f32x4v*>(a.scale + nt0 * 16 + lq * 4), shv = *reinterpret_cast<const f32x4v*>(a.shift + nt0 * 16 + lq * 4);
    int abase[MT], orc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, col = (t & 1) * 16 + lm;
        abase[mt] = row * RS + col * 2 + (lq >> 1) * PLANE + (lq & 1);
        orc[mt] = row | (col << 8);
    }
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)p.out_bytes : 0, 0x00020000);
    const unsigned opix = (unsigned)a.cout * 4u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // (weights and the first patch have landed)
    for (int it = 0; tile < p.ntiles; tile += nwg, ++it) {
        const TilePos here = decode_tile(tile);
        const f32x4v* patch = lds + (it & 1) * BUF;
        // output pixel (2y, 2x) of this lane's M tiles; class (py, px) adds (py, px)
        unsigned obase[MT];
        bool oval[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int y = here.ty0 + (orc[mt] & 255), x = here.tx0 + (orc[mt] >> 8);
            oval[mt] = y < a.Ho && x < a.Wo;
            // (a layer with 8 output channels -- reg2d's last transposed layer in training -- fills half an N tile: the
            //  lanes of channels 8..15 store nothing)
            oval[mt] = oval[mt] && nt0 * 16 + lq * 4 < a.cout;
            obase[mt] = (unsigned)(((((here.b * a.DoF + here.zo) * a.HoF + 2 * y) * a.WoF + 2 * x) * a.cout + nt0 * 16 + lq * 4)) * 4u;
        }
        int wbase = 0;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            // taps of the class, and the patch row / column of its first one: 3x3: 1 / 2 taps from the pixel itself; 5x5: 3 taps from
            // the ring (even outputs) or 2 from the pixel (odd ones)
            const int kh = K5 ? 3 - py : py + 1, kw = K5 ? 3 - px : px + 1;
            const int cbase = K5 ? py * RS + px * 2 : 0;
            f32x4v acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            f32x4v skv[MT];
            unsigned ooff[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                ooff[mt] = oval[mt] ? obase[mt] + (unsigned)(py * a.WoF + px) * opix : 0x80000000u;
                skv[mt] = SKIP ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooff[mt], 0, 0)) : (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
            // K order of the class's packed block: tap-major (dy, then dx), channel-minor
#pragma unroll
            for (int dy = 0; dy < KT; ++dy)
#pragma unroll
                for (int dx = 0; dx < KT; ++dx) {
                    if (dy >= kh || dx >= kw) continue;
                    const int tap = dy * kw + dx;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const f32x4v w = wl[(wbase + tap * NCH + c) * 64 + lane];
                        f32x4v A[MT];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) A[mt] = patch[abase[mt] + c * 2 * PLANE + cbase + dy * RS + dx * 2];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], A[mt][j], acc[mt], 0, 0, 0);
                    }
                }
            wbase += kh * kw * NCH;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4v v = acc[mt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = fmaf(v[j], scv[j], shv[j]);
                    if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                    if (SKIP) v[j] += skv[mt][j];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, ooff[mt], 0, MV_STORE_AUX);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // done reading this patch
        __builtin_amdgcn_s_barrier();
    }
}

#ifdef MVSTER_PROBES   // ping-pong form (variant 7): probe build only
// ------------------------------------------------------------------------------------------------------------------
// Ping-pong form of the persistent kernel (variant 7): a workgroup of EIGHT waves, two per SIMD.  The model fitted to
// conv_pers_kernel's measurements says its steady state is 58-61 % MFMA-busy because a wavefront's
// address arithmetic, DMA issue, barrier wait and epilogue (~1 700 cycles per tile) are not overlapped with MFMAs -- not
// by hipcc inside the wave, and not by a second workgroup on the CU either, which runs in phase with the first.  Here
// the overlap is built into the structure (see the main loop).
// ------------------------------------------------------------------------------------------------------------------
template <int MT, int NT, int KW, int SW, int NCH, int KD, bool WREG, int PF, bool SKIP>
__global__ void __launch_bounds__(512) conv_pp_kernel(ConvArgs a, PersArgs p) {
    using G = PersGeom<MT, KW, SW, KD>;
    constexpr int TY = G::TY, KH = G::KH, PW = G::PW, PH = G::PH, PWH = G::PWH, PLANE = G::PLANE, NBLK = G::NBLK;
    constexpr int NTAP = KD * KH * KW, TAPS2D = KH * KW;
    constexpr int BUF = NCH * 2 * PLANE;                    // float4 per patch buffer
    constexpr int NI = NCH * 2 * NBLK;                      // DMA wave-instructions per tile
    constexpr int NIW = (NI + 3) / 4;
    constexpr int CIN = NCH * 16;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave8 >> 2;                             // which half of the workgroup: its own tiles, its own two patch buffers
    const int wave = wave8 & 3;                             // wave inside the half: M tiles and DMA slots as in conv_pers_kernel
    f32x4v* const lds = reinterpret_cast<f32x4v*>(lds_raw) + grp * 2 * BUF;
    f32x4v* const scratch = reinterpret_cast<f32x4v*>(lds_raw) + 4 * BUF;    // 64 float4: target of the surplus DMA slots
    f32x4v* const wl = scratch + 64;                        // [tap][chunk][nt][lane], shared by both halves (unused with WREG)
    const int lm = lane & 15, lq = lane >> 4;
    const int nt0 = blockIdx.y * NT;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);

    // ---- tile-independent per-lane part of the DMA addresses ------------------------------------------------------
    // instruction i = wave + 4n -> (chunk c, plane pl, block blk); lane -> slot blk*64 + lane of that plane
    // slot -> (patch row, px, quad parity).  dbase = byte offset of the lane's 16 bytes relative to the patch origin
    // (0x80000000 for slots that hold no pixel: stays out of range after adding any tile origin < 2^31);
    // dpos = px | py << 8 | pz << 16 for the border tiles, whose out-of-image lanes are sent out of range one by one.
    unsigned dbase[NIW];
    int dpos[NIW];
#pragma unroll
    for (int n = 0; n < NIW; ++n) {
        const int i = wave + 4 * n;
        const int c = i / (2 * NBLK), r = i - c * 2 * NBLK, pl = r / NBLK, blk = r - pl * NBLK;
        const int s = blk * 64 + lane;
        const int q1 = s & 1;
        int prow, px;
        if (SW == 1) {
            const int pix = s >> 1;
            prow = pix / PW;
            px = pix - prow * PW;
        } else {
            int t = s >> 1;
            const int xh = t % PWH;
            t /= PWH;
            prow = t >> 1;
            px = 2 * xh + (t & 1);
        }
        const bool valid = i < NI && prow < G::ROWS && px < PW;
        const int pz = KD == 1 ? 0 : prow / PH, py = prow - pz * PH;
        dpos[n] = px | (py << 8) | (pz << 16);
        dbase[n] = valid ? (unsigned)(((pz * a.Hi + py) * a.Wi + px) * (CIN * 4) + (c * 16 + pl * 8 + q1 * 4) * 4) : 0x80000000u;
    }

    auto decode_tile = [&](unsigned tile) -> TilePos {
        TilePos t;
        // (branch-free: a select on the divisor being 1 would split the loop body into several basic blocks)
        auto div = [&](unsigned n, int k) -> unsigned { return ((__umulhi(n, p.mul[k]) >> p.shr[k]) & ~p.one[k]) | (n & p.one[k]); };
        unsigned q = div(tile, 0);
        t.tx0 = (int)(tile - q * p.tiles_x) * 32;
        unsigned q2 = div(q, 1);
        t.ty0 = (int)(q - q2 * p.tiles_y) * TY;
        const unsigned q3 = div(q2, 2);
        t.zo = (int)(q2 - q3 * (unsigned)a.Do);
        t.b = (int)q3;
        return t;
    };

    // (branch-free: on interior tiles every lane's bounds test passes; `live` = false sends the whole patch out of range --
    //  used for the DMA slot of a tile that does not exist, so that the loop body stays one basic block)
    auto dma_tile = [&](const TilePos& t, int buf, bool live) {
        const int iz0 = t.zo * a.sd - a.pd[0], iy0 = t.ty0 * SW - a.ph[0], ix0 = t.tx0 * SW - a.pw[0];
        // byte offset of the patch origin (may be negative on border tiles: 32-bit wrap-around arithmetic)
        const unsigned origin = (unsigned)((((t.b * a.Di + iz0) * a.Hi + iy0) * a.Wi + ix0) * (CIN * 4));
        const unsigned wi = live ? (unsigned)a.Wi : 0u;
        f32x4v* const dst0 = lds + buf * BUF;
#pragma unroll
        for (int n = 0; n < NIW; ++n) {
            const int i = wave + 4 * n;
            {
                const int ix = ix0 + (dpos[n] & 255), iy = iy0 + ((dpos[n] >> 8) & 255), iz = iz0 + (dpos[n] >> 16);
                bool ok = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < wi;
                if (KD > 1) ok = ok && (unsigned)iz < (unsigned)a.Di;
                // (named operands: hipcc 7.2 silently drops the kernel's host stub when this builtin is handed an arithmetic
                //  expression as its offset)
                const unsigned off = ok ? dbase[n] + origin : 0x80000000u;
                // (a wave without an n-th slot still issues it -- into the scratch block, every lane out of range: no
                //  wave-dependent branch in the loop body)
                f32x4v* const dst = (NI % 4 == 0 || n + 1 < NIW || i < NI) ? dst0 + (i / NBLK) * PLANE + (i % NBLK) * 64 : scratch;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
            }
        }
    };

    // ---- once per workgroup: weights, scale / shift, A-read bases ------------------------------------------------
    const unsigned nwg = gridDim.x;
    const unsigned first = xcd_remap(blockIdx.x, nwg);   // the workgroup's tiles: first, first + nwg, ...; half grp takes every other one
    const unsigned ntl = first < p.ntiles ? (p.ntiles - first + nwg - 1) / nwg : 0u;       // tiles of this workgroup
    const int kcount = (int)((ntl + 1u - (unsigned)grp) >> 1);                           // ... of this half
    auto tile_of = [&](int k) -> unsigned { return first + (unsigned)(2 * k + grp) * nwg; };
    if (kcount > 0) dma_tile(decode_tile(tile_of(0)), 0, true);
    if (grp == 0 && kcount > 1) dma_tile(decode_tile(tile_of(1)), 1, true);   // (half 0 computes first: its second patch too)
    f32x4v wreg[WREG ? NTAP * NCH * NT : 1];
    const long wstep = (long)a.ntile_total * 256;          // floats per K step of the packed weights
    if (WREG) {
#pragma unroll
        for (int s = 0; s < NTAP * NCH; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wreg[s * NT + nt] = *reinterpret_cast<const f32x4v*>(a.wpk + s * wstep + ((long)(nt0 + nt) * 64 + lane) * 4);
    } else {
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.wpk), (short)0, (int)(NTAP * NCH * wstep * 4), 0x00020000);
        for (int i = wave8; i < NTAP * NCH * NT; i += 8) {
            const int s = i / NT, nt = i - s * NT;
            const unsigned off = (unsigned)((s * wstep + (long)(nt0 + nt) * 256) * 4) + lane * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(wl + i * 64), 16, off, 0, 0, 0);
        }
    }
    f32x4v scv[NT], shv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = (nt0 + nt) * 16 + lq * 4;
        scv[nt] = *reinterpret_cast<const f32x4v*>(a.scale + n0);
        shv[nt] = *reinterpret_cast<const f32x4v*>(a.shift + n0);
    }
    // float4 index (inside a chunk's plane pair) of this lane's A operand for tap (0,0,0) of each of its M tiles
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, x = (t & 1) * 16 + lm;
        const int rs = G::ROWSLOTS;
        abase[mt] = (SW == 1 ? row * rs + x * 2 : row * 2 * rs + x * 2) + (lq >> 1) * PLANE + (lq & 1);
    }
    // output (and skip) byte offset of this lane's 4 channels of each M tile, relative to the tile's first pixel
    const __amdgpu_buffer_rsrc_t out_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(SKIP ? a.skip : a.in), (short)0, SKIP ? (int)p.out_bytes : 0, 0x00020000);
    unsigned obase[MT];
    int orc[MT];            // row | col << 8 of the M tile's pixel inside the workgroup tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int t = wave * MT + mt;
        const int row = t >> 1, col = (t & 1) * 16 + lm;
        orc[mt] = row | (col << 8);
        obase[mt] = (unsigned)((row * a.Wo + col) * a.cout + nt0 * 16 + lq * 4) * 4u;
    }
    auto tap_off = [&](int kz, int ky, int kx) -> int {     // float4 offset of a tap relative to abase (compile-time)
        const int prow = kz * PH + ky;
        return SW == 1 ? prow * G::ROWSLOTS + kx * 2 : prow * G::ROWSLOTS + (kx & 1) * PWH * 2 + (kx >> 1) * 2;
    };

    __syncthreads();        // (waits vmcnt(0): first patch and the weights have landed)

    // The loop body is ONE basic block (2-D kernels): the DMA of tile t+1, the MFMAs of tile t and the epilogue of tile
    // t-1 are independent instruction streams that the scheduler interleaves -- at one wavefront per SIMD nothing else
    // could fill the matrix pipe's shadow.  Stores stay below the DMA issues (sched_barrier) so that a counted
    // vmcnt(stores) before the barrier means "the next patch has landed" without draining the stores.
    f32x4v pacc[MT][NT], pskv[MT][NT];
    unsigned pooff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        pooff[mt] = 0x80000000u;                 // nothing to store in the first round
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pacc[mt][nt] = pskv[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    }
    auto epilogue = [&](const f32x4v (&accv)[MT][NT], const f32x4v (&skvv)[MT][NT], const unsigned (&off)[MT]) {
        // the accumulator is D^T (weights in the A slot): 4 consecutive output channels of one voxel per lane
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4v v = accv[mt][nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = fmaf(v[j], scv[nt][j], shv[nt][j]);
                    if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                    if (SKIP) v[j] += skvv[mt][nt][j];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, off[mt] + nt * 64, 0, MV_STORE_AUX);
            }
        }
    };

    // Half-phases: in even ones half 0 runs the MFMAs of its tile k while half 1 "prepares" (epilogue of its previous tile,
    // output offsets / skip loads of its next one, DMA issue for the one after, counted wait for the patch it is about to
    // use); in odd ones the roles swap; one barrier of all eight waves per half-phase.  The two wavefronts that share a SIMD
    // are therefore ALWAYS in anti-phase: one feeds the matrix pipe, the other issues its scalar / vector / memory work in
    // the pipe's shadow.
    f32x4v skvN[MT][NT];
    unsigned ooffN[MT];
    TilePos here = decode_tile(kcount > 0 ? tile_of(0) : 0u);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ooffN[mt] = 0x80000000u;
    const int nhalf = 2 * (int)((ntl + 1u) >> 1) + 1;
    // (the prepare step of half 0 belongs to odd half-phases, that of half 1 to even ones; both start by preparing tile 0's
    //  offsets here, outside the loop, because half 0's first MFMA phase comes before its first prepare phase)
    {
        const unsigned oorigin = (unsigned)((((here.b * a.Do + here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool ok = grp == 0 && kcount > 0 && here.ty0 + (orc[mt] & 255) < a.Ho && here.tx0 + (orc[mt] >> 8) < a.Wo;
            ooffN[mt] = ok ? obase[mt] + oorigin : 0x80000000u;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                skvN[mt][nt] = SKIP ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooffN[mt] + nt * 64, 0, 0))
                                    : (f32x4v){0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int hp = 0; hp < nhalf; ++hp) {
        MV_PPTL(0, __builtin_amdgcn_s_memtime());
        MV_PPTL(3, __builtin_amdgcn_s_getreg(63492));
        if ((hp & 1) == grp) {
            // ---------------- MFMA half-phase: tile k of this half --------------------------------------------------
            const int k = (hp - grp) >> 1;
            if (k < kcount) {
                const int cur = k & 1;
                f32x4v acc[MT][NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
                const f32x4v* patch = lds + cur * BUF;
                const int kz_lo = KD == 1 ? 0 : max(0, a.pd[0] - here.zo * a.sd), kz_hi = KD == 1 ? 1 : min(KD, a.Di + a.pd[0] - here.zo * a.sd);
#pragma unroll
                for (int kz = 0; kz < KD; ++kz) {
                    if (kz < kz_lo || kz >= kz_hi) continue;
                    f32x4v A[PF + 1][NCH][MT], Bv[PF + 1][NCH][NT];
                    auto load_tap = [&](int t2, f32x4v (&Aa)[NCH][MT], f32x4v (&Bb)[NCH][NT]) {
                        const int ky = t2 / KW, kx = t2 - ky * KW;
                        const int to = tap_off(kz, ky, kx);
                        const int tap = kz * TAPS2D + t2;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) Aa[c][mt] = patch[abase[mt] + c * 2 * PLANE + to];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                Bb[c][nt] = WREG ? wreg[(tap * NCH + c) * NT + nt] : wl[((tap * NCH + c) * NT + nt) * 64 + lane];
                        }
                    };
#pragma unroll
                    for (int t2 = 0; t2 < PF && t2 < TAPS2D; ++t2) load_tap(t2, A[t2 % (PF + 1)], Bv[t2 % (PF + 1)]);
#pragma unroll
                    for (int t2 = 0; t2 < TAPS2D; ++t2) {
                        if (t2 + PF < TAPS2D) load_tap(t2 + PF, A[(t2 + PF) % (PF + 1)], Bv[(t2 + PF) % (PF + 1)]);
#pragma unroll
                        for (int c = 0; c < NCH; ++c)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt)
                                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bv[t2 % (PF + 1)][c][nt][j], A[t2 % (PF + 1)][c][mt][j],
                                                                                           acc[mt][nt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    pooff[mt] = ooffN[mt];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        pacc[mt][nt] = acc[mt][nt];
                        pskv[mt][nt] = skvN[mt][nt];
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): done reading this patch
            MV_PPTL(2, k < kcount ? 1 : 0);
        } else {
            // ---------------- prepare half-phase (the other half owns the matrix pipe) --------------------------------
            const int kn = (hp + 1 - grp) >> 1;                          // the tile whose MFMAs come next half-phase
            epilogue(pacc, pskv, pooff);                                 // tile kn - 1 (out-of-range offsets: nothing stored)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) pooff[mt] = 0x80000000u;
            __builtin_amdgcn_sched_barrier(0);
            const bool have = kn < kcount;
            here = decode_tile(have ? tile_of(kn) : 0u);
            const unsigned oorigin = (unsigned)((((here.b * a.Do + here.zo) * a.Ho + here.ty0) * a.Wo + here.tx0) * a.cout) * 4u;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bool ok = have && here.ty0 + (orc[mt] & 255) < a.Ho && here.tx0 + (orc[mt] >> 8) < a.Wo;
                ooffN[mt] = ok ? obase[mt] + oorigin : 0x80000000u;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    skvN[mt][nt] = SKIP ? __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ooffN[mt] + nt * 64, 0, 0))
                                        : (f32x4v){0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool more = kn + 1 < kcount;
            dma_tile(decode_tile(more ? tile_of(kn + 1) : 0u), (kn + 1) & 1, more);
            // the patch of tile kn (its DMA was issued one prepare phase ago) has landed once only this phase's own
            // memory operations are outstanding: MT*NT stores, the skip loads, NIW DMA instructions
            MV_PPTL(2, __builtin_amdgcn_s_memtime());      // (probe build: prepare work issued, before the wait)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MT * NT * (SKIP ? 2 : 1) + NIW) : "memory");
        }
        MV_PPTL(1, __builtin_amdgcn_s_memtime());
        __builtin_amdgcn_s_barrier();
    }
}
#endif  // MVSTER_PROBES

// ------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions with many output channels (variant 6): the 64 -> 144 / 64 -> 72 "tap" convolutions of the
// re-associated FPN levels (conv_plan.FpnPlan), 64 -> 64 / 32 -> 64 laterals.  The direct kernel walks 3 (5) N tiles per
// pass over the input, i.e. reads the input 3 (1) times and the weights from L1 for every M tile: 135.6 MB of HBM traffic
// per launch against 85 MB of input + output (profiles/r02_b_pmc_summary.txt).  Here a persistent workgroup keeps ALL
// packed weights in LDS (36 KB for 64 -> 144, fetched once by LDS-DMA), a wavefront holds the K = cin operands of its
// MT M tiles in registers (read once, prefetched one group ahead) and loops over the N tiles: B fragments from LDS,
// 16*MT MFMAs, fused epilogue, float4 store.
// ------------------------------------------------------------------------------------------------------------------
template <int NCH, int MT>
__global__ void __launch_bounds__(256) conv1x1_pers_kernel(ConvArgs a, unsigned ngroups, unsigned mtot, unsigned out_bytes) {
    constexpr int CIN = NCH * 16;
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    f32x4v* const wl = reinterpret_cast<f32x4v*>(lds_raw);            // [K step][N tile][lane], the packed array verbatim
    const int ntile = a.ntile_total;
    f32x4v* const ssl = wl + NCH * ntile * 64;                        // scale [ntile*4] float4, then shift [ntile*4]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lm = lane & 15, lq = lane >> 4;
    const __amdgpu_buffer_rsrc_t in_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), (short)0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (int)out_bytes, 0x00020000);
    // skip_mode 1: same-shape tensor; 2: [B, 1, Ho/2, Wo/2, cout] map, added through a bilinear x2 (align_corners) up-sampling
    const __amdgpu_buffer_rsrc_t skip_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.skip_mode ? a.skip : a.in), (short)0,
        a.skip_mode == 1 ? (int)out_bytes : a.skip_mode == 2 ? (int)(out_bytes / 4) : 0, 0x00020000);
    {
        const __amdgpu_buffer_rsrc_t w_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wpk), (short)0, NCH * ntile * 1024, 0x00020000);
        for (int i = wave; i < NCH * ntile; i += 4) {
            const unsigned off = (unsigned)(i * 1024 + lane * 16);
            f32x4v* const dst = wl + i * 64;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)dst, 16, off, 0, 0, 0);
        }
        for (int i = threadIdx.x; i < ntile * 4; i += 256) {
            ssl[i] = *reinterpret_cast<const f32x4v*>(a.scale + i * 4);
            ssl[ntile * 4 + i] = *reinterpret_cast<const f32x4v*>(a.shift + i * 4);
        }
    }
    auto load_group = [&](unsigned g, f32x4v (&A)[MT][NCH], unsigned (&vox)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned m = ((g * 4 + wave) * MT + mt) * 16 + lm;
            vox[mt] = m;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const unsigned off = m < mtot ? (m * CIN + c * 16 + lq * 4) * 4u : 0x80000000u;
                A[mt][c] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0));
            }
        }
    };
    f32x4v A[MT][NCH], An[MT][NCH];
    unsigned vox[MT], voxn[MT];
    struct Up {                 // skip_mode 2: the four half-resolution texels (byte offsets) and weights of an output voxel
        unsigned o00, o01, o10, o11;
        mv::Lerp ly, lx;
    } up[MT];
    unsigned g = blockIdx.x;
    if (g < ngroups) load_group(g, A, vox);
    __syncthreads();            // weights (LDS-DMA: vmcnt(0)) and scale / shift in LDS
    const unsigned cout4 = (unsigned)a.cout * 4u;
    for (; g < ngroups; g += gridDim.x) {
        const unsigned gn = g + gridDim.x < ngroups ? g + gridDim.x : g;
        load_group(gn, An, voxn);                                     // next group's operands fly under this group's math
        if (a.skip_mode == 2) {
            const int hh = a.HoF / 2, wh = a.WoF / 2;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned m = vox[mt] < mtot ? vox[mt] : 0u;
                const unsigned t = fast_div(m, (unsigned)a.Wo, a.div_mul[0], a.div_shr[0]);
                const int ox = (int)(m - t * (unsigned)a.Wo);
                const unsigned b = fast_div(t, (unsigned)a.Ho, a.div_mul[1], a.div_shr[1]);
                const int oy = (int)(t - b * (unsigned)a.Ho);
                up[mt].ly = mv::make_lerp(oy, hh, a.HoF);
                up[mt].lx = mv::make_lerp(ox, wh, a.WoF);
                const unsigned base = b * (unsigned)(hh * wh);
                up[mt].o00 = (base + up[mt].ly.i0 * wh + up[mt].lx.i0) * cout4;
                up[mt].o01 = (base + up[mt].ly.i0 * wh + up[mt].lx.i1) * cout4;
                up[mt].o10 = (base + up[mt].ly.i1 * wh + up[mt].lx.i0) * cout4;
                up[mt].o11 = (base + up[mt].ly.i1 * wh + up[mt].lx.i1) * cout4;
            }
        }
        f32x4v Bv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) Bv[c] = wl[(c * ntile) * 64 + lane];
        for (int nt = 0; nt < ntile; ++nt) {
            f32x4v Bn[NCH];
            const int ntn = nt + 1 < ntile ? nt + 1 : nt;
#pragma unroll
            for (int c = 0; c < NCH; ++c) Bn[c] = wl[(c * ntile + ntn) * 64 + lane];
            const f32x4v sc = ssl[nt * 4 + lq], sh = ssl[ntile * 4 + nt * 4 + lq];
            f32x4v acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Bv[c][j], A[mt][c][j], acc[mt], 0, 0, 0);
            const unsigned n0 = (unsigned)(nt * 16 + lq * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned off = (n0 < (unsigned)a.cout && vox[mt] < mtot) ? vox[mt] * cout4 + n0 * 4u : 0x80000000u;
                f32x4v v = acc[mt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = fmaf(v[j], sc[j], sh[j]);
                    if (a.relu) v[j] = fmaxf(v[j], 0.0f);
                }
                if (a.skip_mode == 1) {
                    const f32x4v k = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, off, 0, 0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += k[j];
                } else if (a.skip_mode == 2) {
                    // F.interpolate(prev, x2, bilinear, align_corners) + inner(conv) (mvs4net_utils.py:482-488), the direct
                    // kernel's arithmetic (epilogue_store in conv_mfma.hip)
                    const bool ok = n0 < (unsigned)a.cout && vox[mt] < mtot;
                    const unsigned c4 = n0 * 4u;
                    const f32x4v k00 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ok ? up[mt].o00 + c4 : 0x80000000u, 0, 0));
                    const f32x4v k01 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ok ? up[mt].o01 + c4 : 0x80000000u, 0, 0));
                    const f32x4v k10 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ok ? up[mt].o10 + c4 : 0x80000000u, 0, 0));
                    const f32x4v k11 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(skip_rsrc, ok ? up[mt].o11 + c4 : 0x80000000u, 0, 0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = mv::bilerp(up[mt].ly, up[mt].lx, k00[j], k01[j], k10[j], k11[j]) + v[j];
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), out_rsrc, off, 0, MV_STORE_AUX);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) Bv[c] = Bn[c];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            vox[mt] = voxn[mt];
#pragma unroll
            for (int c = 0; c < NCH; ++c) A[mt][c] = An[mt][c];
        }
    }
}


template <int MT, int NT, int KW, int SW, int NCH, int KD, bool WREG, int PF, bool SKIP, bool LD>
int launch_pers(const ConvArgs& a, int wpc, hipStream_t s) {
    using G = PersGeom<MT, KW, SW, KD>;
    constexpr int NTAP = KD * KW * KW;
    const size_t lds = (size_t)(2 * NCH * 2 * G::PLANE + 64 + (WREG ? 0 : NTAP * NCH * NT * 64)) * 16;
    if (lds > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv_pers_kernel<MT, NT, KW, SW, NCH, KD, WREG, PF, SKIP, LD>;
    static unsigned long attr_done = 0;
    if (lds > 64 * 1024 && !allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int g_num_cu = num_cus();
    if (g_num_cu <= 0) return MVSTER_ERR_LAUNCH;
    PersArgs p;
    if (!fill_pers_args(a, G::TY, p)) return MVSTER_ERR_UNSUPPORTED;
    const long ntiles = p.ntiles;
    p.prio = (wpc >> 4) & 1;
    wpc &= 15;
    // resident workgroups per CU: what LDS allows, at most 4 (registers), unless the caller pins it
    int by_lds = (int)((160 * 1024) / lds);
    int per_cu = wpc > 0 ? wpc : (by_lds > 3 ? 3 : by_lds);
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu < 1) per_cu = 1;
    const int ny = a.ntile_total / NT;
    long gmax = (long)g_num_cu * per_cu / ny;
    if (gmax < 1) gmax = 1;
    // equal shares: every workgroup walks the same number of tiles (no straggler round)
    const long rounds = (ntiles + gmax - 1) / gmax;
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv_pers_kernel<%d, %d, %d, %d, %d, %d, %s, %d, %s, %s>", MT, NT, KW, SW, NCH, KD, WREG ? "true" : "false", PF,
                   SKIP ? "true" : "false", LD ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, ny, 1), dim3(LD ? 512 : 256), lds, s, a, p);
    return mv_check_launch();
}

template <int NCH, bool SKIP, int KS>
int launch_tpers(const ConvArgs& a, int wpc, hipStream_t s) {
    using G = PersGeom<2, KS == 5 ? 3 : 2, 1, 1>;
    const size_t lds = (size_t)(2 * NCH * 2 * G::PLANE + 64 + (KS == 5 ? 25 : 9) * NCH * 64) * 16;
    if (lds > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv_tpers_kernel<NCH, SKIP, KS>;
    static unsigned long attr_done = 0;
    if (lds > 64 * 1024 && !allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    PersArgs p;
    if (!fill_pers_args(a, G::TY, p)) return MVSTER_ERR_UNSUPPORTED;          // (tiles over the INPUT lattice: a.Ho, a.Wo, a.Do)
    const long ntiles = p.ntiles;
    wpc &= 15;
    const int by_lds = (int)((160 * 1024) / lds);
    int per_cu = wpc > 0 ? wpc : 2;
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu > 2) per_cu = 2;
    long gmax = (long)ncu * per_cu / a.ntile_total;
    if (gmax < 1) gmax = 1;
    const long rounds = (ntiles + gmax - 1) / gmax;
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv_tpers_kernel<%d, %s, %d>", NCH, SKIP ? "true" : "false", KS);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, a.ntile_total, 1), dim3(512), lds, s, a, p);
    return mv_check_launch();
}

template <int KW, bool SKIP>
int launch_pers8(const ConvArgs& a, int wpc, hipStream_t s) {
    using G = PersGeom<2, KW, 2, 1>;
    const size_t lds = (size_t)(2 * G::PLANE + 64) * 16;
    auto kern = conv_pers8_kernel<KW, SKIP>;
    static unsigned long attr_done = 0;
    if (lds > 64 * 1024 && !allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int ncu = num_cus();
    if (ncu <= 0) return MVSTER_ERR_LAUNCH;
    PersArgs p;
    if (!fill_pers_args(a, G::TY, p)) return MVSTER_ERR_UNSUPPORTED;
    const long ntiles = p.ntiles;
    wpc &= 15;
    const int by_lds = (int)((160 * 1024) / lds);
    int per_cu = wpc > 0 ? wpc : 2;
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu > 3) per_cu = 3;
    const long gmax = (long)ncu * per_cu;
    const long rounds = (ntiles + gmax - 1) / gmax;       // equal shares
    const long gx = (ntiles + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv_pers8_kernel<%d, %s>", KW, SKIP ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, 1, 1), dim3(512), lds, s, a, p);
    return mv_check_launch();
}

#ifdef MVSTER_PROBES
template <int MT, int NT, int KW, int SW, int NCH, int KD, bool WREG, int PF, bool SKIP>
int launch_pp(const ConvArgs& a, hipStream_t s) {
    using G = PersGeom<MT, KW, SW, KD>;
    constexpr int NTAP = KD * KW * KW;
    const size_t lds = (size_t)(4 * NCH * 2 * G::PLANE + 64 + (WREG ? 0 : NTAP * NCH * NT * 64)) * 16;
    if (lds > 160 * 1024) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv_pp_kernel<MT, NT, KW, SW, NCH, KD, WREG, PF, SKIP>;
    static unsigned long attr_done = 0;
    if (!allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int g_num_cu = num_cus();
    if (g_num_cu <= 0) return MVSTER_ERR_LAUNCH;
    PersArgs p;
    if (!fill_pers_args(a, G::TY, p)) return MVSTER_ERR_UNSUPPORTED;
    const long ntiles = p.ntiles;
    // one workgroup (8 waves) per CU; every workgroup walks the same number of tiles, an even number where possible
    // (both halves busy in every half-phase)
    const int ny = a.ntile_total / NT;
    long gmax = (long)g_num_cu / ny;
    if (gmax < 1) gmax = 1;
    long per = (ntiles + gmax - 1) / gmax;
    if (per > 1 && (per & 1)) ++per;
    const long gx = (ntiles + per - 1) / per;
    MV_NOTE_KERNEL("conv_pp_kernel<%d, %d, %d, %d, %d, %d, %s, %d, %s>", MT, NT, KW, SW, NCH, KD, WREG ? "true" : "false", PF,
                   SKIP ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, ny, 1), dim3(512), lds, s, a, p);
    return mv_check_launch();
}
#endif  // MVSTER_PROBES

}  // namespace

#ifdef MVSTER_TIMELINE
extern "C" int mvster_debug_pers_timeline(void* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ptl), &buf, sizeof(buf)) == hipSuccess ? MVSTER_OK : MVSTER_ERR_LAUNCH;
}
#endif

namespace {
template <int NCH, int MT>
int launch_1x1(const ConvArgs& a, int wpc, hipStream_t s) {
    const size_t lds = (size_t)(NCH * a.ntile_total * 64 + 2 * a.ntile_total * 4) * 16;
    const long mtot = (long)a.B * a.Do * a.Ho * a.Wo;
    const long out_bytes = mtot * a.cout * 4;
    if (lds > 160 * 1024 || a.in_bytes >= (1u << 31) || out_bytes >= (1L << 31) || a.cout % 4 != 0) return MVSTER_ERR_UNSUPPORTED;
    auto kern = conv1x1_pers_kernel<NCH, MT>;
    static unsigned long attr_done = 0;
    if (!allow_big_lds(reinterpret_cast<const void*>(kern), attr_done)) return MVSTER_ERR_LAUNCH;
    const int g_num_cu = num_cus();
    if (g_num_cu <= 0) return MVSTER_ERR_LAUNCH;
    const long ngroups = (mtot + 64 * MT - 1) / (64 * MT);
    int by_lds = (int)((160 * 1024) / lds);
    int per_cu = wpc > 0 ? wpc : 2;
    if (per_cu > by_lds) per_cu = by_lds;
    if (per_cu > 4) per_cu = 4;
    const long gmax = (long)g_num_cu * per_cu;
    const long rounds = (ngroups + gmax - 1) / gmax;
    const long gx = (ngroups + rounds - 1) / rounds;
    MV_NOTE_KERNEL("conv1x1_pers_kernel<%d, %d>", NCH, MT);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, s, a, (unsigned)ngroups, (unsigned)mtot, (unsigned)out_bytes);
    return mv_check_launch();
}
}  // namespace

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n = prop.multiProcessorCount;
        // (probe build: MVSTER_CUS = the CU count the persistent grids are sized for -- fewer workgroups per launch leave room
        //  for the kernels of the other depth map in flight; the product library reads no environment)
        if (const char* e = MV_PROBE_ENV("MVSTER_CUS")) {
            const int v = atoi(e);
            if (v > 0 && v <= n) n = v;
        }
    }
    return n;
}

// 1x1x1 stride-1 convolutions, cin in {32, 64}, any cout % 4 == 0, optional same-shape skip (variant 6).
int dispatch_1x1(const ConvArgs& a, int mt, int wpc, hipStream_t s) {
    if (a.nclass != 1 || a.kd[0] != 1 || a.kh[0] != 1 || a.kw[0] != 1 || a.sd != 1 || a.sh != 1 || a.sw != 1 || a.osd != 1 ||
        a.osh != 1 || a.osw != 1 || a.pd[0] != 0 || a.ph[0] != 0 || a.pw[0] != 0 || a.skip_mode > 2 || a.prob_w)
        return MVSTER_ERR_UNSUPPORTED;
    if (a.skip_mode == 2 && (a.Do != 1 || (a.HoF & 1) || (a.WoF & 1))) return MVSTER_ERR_UNSUPPORTED;      // (2-D, even sizes)
    wpc &= 15;
    if (a.cin == 64 && mt == 1) return launch_1x1<4, 1>(a, wpc, s);
    if (a.cin == 64 && mt == 2) return launch_1x1<4, 2>(a, wpc, s);
    if (a.cin == 32 && mt == 1) return launch_1x1<2, 1>(a, wpc, s);
    if (a.cin == 32 && mt == 2) return launch_1x1<2, 2>(a, wpc, s);
    if (a.cin == 32 && mt == 4) return launch_1x1<2, 4>(a, wpc, s);
    if (a.cin == 16 && mt == 2) return launch_1x1<1, 2>(a, wpc, s);
    if (a.cin == 16 && mt == 4) return launch_1x1<1, 4>(a, wpc, s);
    return MVSTER_ERR_UNSUPPORTED;
}

// variant 7: the ping-pong form (probe build only: measured no faster than variant 5)
#ifndef MVSTER_PROBES
int dispatch_pp(const ConvArgs&, int, int, hipStream_t) { return MVSTER_ERR_UNSUPPORTED; }
#else
// variant 7: the ping-pong form; the instances whose four patch buffers (+ resident weights) fit 160 KB of LDS
int dispatch_pp(const ConvArgs& a, int mt, int nt, hipStream_t s) {
    if (a.nclass != 1 || a.osd != 1 || a.osh != 1 || a.osw != 1 || a.skip_mode > 1 || a.prob_w || a.cout % 16 != 0 ||
        a.sh != a.sw || a.kh[0] != a.kw[0] || a.ntile_total % nt != 0 || mt != 2 || a.cin % 16 != 0)
        return MVSTER_ERR_UNSUPPORTED;
    const int kd = a.kd[0], kw = a.kw[0], sw = a.sw, nch = a.cin / 16;
#define MV_PP(NT_, KW_, SW_, NCH_, KD_, WREG_, PF_) \
    if (nt == NT_ && kw == KW_ && sw == SW_ && nch == NCH_ && kd == KD_)                                        \
        return a.skip_mode == 1 ? launch_pp<2, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_, true>(a, s)                 \
                                : launch_pp<2, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_, false>(a, s);
    MV_PP(1, 3, 1, 1, 1, true, 2)      // 16 -> 16 3x3
    MV_PP(2, 3, 1, 2, 1, false, 1)     // 32 -> 32 3x3
    MV_PP(2, 3, 2, 1, 1, true, 1)      // 16 -> 32 3x3 stride 2
#undef MV_PP
    return MVSTER_ERR_UNSUPPORTED;
}
#endif  // MVSTER_PROBES

// Layers the family covers: ordinary (non-transposed) convolutions, cin in {16, 32, 64}, cout % 16 == 0, kernel (1|3) x 3 x 3
// or 1 x 5 x 5 with "same" padding geometry handled by the generic bounds checks, stride 1 or 2 in-plane.
int dispatch_pers(const ConvArgs& a, int mt, int nt, int wpc, hipStream_t s) {
    if (a.nclass == 4 && a.osd == 1 && a.osh == 2 && a.osw == 2 && a.sd == 1 && a.sh == 1 && a.sw == 1 && a.skip_mode <= 1 &&
        !a.prob_w && (a.cout % 16 == 0 || a.cout == 8) && (a.cin == 16 || a.cin == 32 || a.cin == 64) && mt == 2 && nt == 1) {
        // transposed 1x3x3 stride (1,2,2): the classes must be the 1 / 2 / 2 / 4-tap ones in (py, px) order, unpadded;
        // 1x5x5: the 9 / 6 / 6 / 4-tap ones, an even parity padded by one (16 / 32 input channels: 25 KB of weights per chunk)
        const bool k5 = a.kh[0] == 3;
        if (k5 && a.cin == 64) return MVSTER_ERR_UNSUPPORTED;
        for (int c = 0; c < 4; ++c) {
            const int py = c >> 1, px = c & 1;
            const int kh = k5 ? 3 - py : py + 1, kw = k5 ? 3 - px : px + 1, ph = k5 ? 1 - py : 0, pw = k5 ? 1 - px : 0;
            if (a.kd[c] != 1 || a.kh[c] != kh || a.kw[c] != kw || a.pd[c] || a.ph[c] != ph || a.pw[c] != pw || a.od[c] ||
                a.oh[c] != py || a.ow[c] != px || a.nsteps[c] != kh * kw * (a.cin / 16))
                return MVSTER_ERR_UNSUPPORTED;
        }
#define MV_TP(NCH_)                                                                                                   \
    return k5 ? (a.skip_mode == 1 ? launch_tpers<NCH_, true, 5>(a, wpc, s) : launch_tpers<NCH_, false, 5>(a, wpc, s)) \
              : (a.skip_mode == 1 ? launch_tpers<NCH_, true, 3>(a, wpc, s) : launch_tpers<NCH_, false, 3>(a, wpc, s));
        if (a.cin == 16) { MV_TP(1) }
        if (a.cin == 32) { MV_TP(2) }
#undef MV_TP
        return a.skip_mode == 1 ? launch_tpers<4, true, 3>(a, wpc, s) : launch_tpers<4, false, 3>(a, wpc, s);
    }
    if (a.nclass != 1 || a.osd != 1 || a.osh != 1 || a.osw != 1 || a.skip_mode > 1 || a.prob_w || a.cout % 16 != 0 ||
        a.sh != a.sw || a.kh[0] != a.kw[0] || a.ntile_total % nt != 0 || mt != 2)
        return MVSTER_ERR_UNSUPPORTED;
    const int kd = a.kd[0], kw = a.kw[0], sw = a.sw, nch = a.cin / 16;
    if (a.cin == 8 && a.cout == 16 && sw == 2 && kd == 1 && a.sd == 1 && nt == 1 && (kw == 3 || kw == 5) && a.ph[0] == kw / 2 &&
        a.pw[0] == kw / 2) {
        // 8 -> 16 channels, stride 2: conv_pers8_kernel
        if (kw == 5) return a.skip_mode == 1 ? launch_pers8<5, true>(a, wpc, s) : launch_pers8<5, false>(a, wpc, s);
        return a.skip_mode == 1 ? launch_pers8<3, true>(a, wpc, s) : launch_pers8<3, false>(a, wpc, s);
    }
    if (a.cin % 16 != 0) return MVSTER_ERR_UNSUPPORTED;
#define MV_P(NT_, KW_, SW_, NCH_, KD_, WREG_, PF_) MV_Q(2, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_)
    const bool ld = (wpc & 32) != 0;          // bit 5 of the workgroups-per-CU field: waves 4-7 issue the DMA
#define MV_Q(MT_, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_) \
    if (mt == MT_ && nt == NT_ && kw == KW_ && sw == SW_ && nch == NCH_ && kd == KD_) {                            \
        if (ld) {                                                                                                  \
            /* (built for the stride-2 families only: on the stride-1 ones the Winograd kernels are ahead anyway) */  \
            if constexpr (SW_ == 2)                                                                                \
                return a.skip_mode == 1 ? launch_pers<MT_, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_, true, true>(a, wpc, s)   \
                                        : launch_pers<MT_, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_, false, true>(a, wpc, s); \
            return MVSTER_ERR_UNSUPPORTED;                                                                         \
        }                                                                                                          \
        return a.skip_mode == 1 ? launch_pers<MT_, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_, true, false>(a, wpc, s)   \
                                : launch_pers<MT_, NT_, KW_, SW_, NCH_, KD_, WREG_, PF_, false, false>(a, wpc, s);  \
    }
    MV_P(1, 3, 1, 1, 1, true, 2)      // 16 -> 16 3x3           (FPN conv1.1/1.2, composed mid level)
    MV_P(2, 3, 1, 2, 1, false, 1)     // 32 -> 32 3x3           (FPN conv2.1/2.2)
    MV_P(1, 3, 1, 2, 1, false, 1)     // 32 -> N 3x3, one N tile per workgroup
    MV_P(1, 3, 1, 4, 1, false, 1)     // 64 -> N 3x3, one N tile per workgroup (36 KB of weights + 2 x 58 KB of patch: one workgroup per CU)
    MV_P(2, 5, 2, 1, 1, false, 1)     // 16 -> 32 5x5 stride 2  (FPN conv2.0)
    MV_P(1, 3, 1, 1, 3, false, 2)     // 16 -> 16 3x3x3         (reg2d conv2)
    MV_P(2, 3, 2, 1, 1, false, 1)     // 16 -> 32 3x3 stride 2  (reg2d conv3)
    // (8-row tiles, mt = 4, were measured no faster on any layer: 26.8 vs 26.0 us and 45.1 vs 45.0 us on the two largest)
#undef MV_P
#undef MV_Q
    return MVSTER_ERR_UNSUPPORTED;
}

}  // namespace mvconv
