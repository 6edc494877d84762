// Training-mode BatchNorm + ReLU on channels-last activations [rows, C] (C in {4,...,64}, multiple of 4):
// the elementwise half of the reference's conv -> BatchNorm -> ReLU blocks (models/mvs4net_utils.py:116-123,
// :224-251) under autograd.  HBM-bound streaming kernels.  Forward 2 reads + 1 write; backward 4 reads + 1 write
// (PyTorch's autograd over the unfused ops makes ~19 passes).
//   y  = relu(x * scale + shift)                  scale = gamma * rstd, shift = beta - mean * scale
//   g  = gy * (y > 0)            xh = (x - mean) * rstd
//   dbeta = sum g     dgamma = sum g * xh         dx = scale * (g - dbeta/N - xh * dgamma/N)
// Every thread owns one float4 column group (blockDim*4 is a multiple of C), so per-channel sums stay in
// registers along the grid-stride loop and are reduced once per workgroup (wave shuffles, then LDS) into a
// per-workgroup slot of `partial` [groups][nblk][2][C].  The LAST workgroup to arrive (a ticket counter) adds the slots in
// a fixed order and in fp64 (deterministic) and writes the finished statistics / sums: one launch per reduction.
// Rounds 1-5 finished in a second launch (7 / 6 us each, 108 launches per training step), because the ticket form written
// with __threadfence() costs an agent-scope release per workgroup, which on gfx950 writes the XCD's L2 back: +5 ms per
// step.  The form here publishes the slots with write-through (sc1) stores and reads them with sc1 loads -- no L2
// write-back, no L1 invalidate (MI355X_MICROARCH.md, "inter-workgroup visibility": sc1 stores and loads on both sides).
#include "common.hpp"

namespace {

__device__ __forceinline__ float bn_act(float x, float sc, float sh) { return fmaf(x, sc, sh); }

// 1 where the gradient passes the ReLU.  Kept as a multiplicative mask: hipcc 7.2 if-converts
// `cond ? g : 0.0f` on a just-loaded g into "g = 0; if (cond) {}" in the apply kernel (wrong code, caught by
// tests/test_gpu_train.py::test_batch_norm_cl_matches_torch).
__device__ __forceinline__ float relu_mask(int relu, float act) { return (relu == 0 || act > 0.0f) ? 1.0f : 0.0f; }

// ---- one-launch reductions: write-through slots, ticket, last arriver finishes -----------------------------------------
constexpr int kRedThreads = 1024;          // workgroup size of the two reduction kernels (<= 256 of them per launch)

__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// True (in every thread) for the last workgroup of the grid to get here.  Every workgroup's st_agent stores issued before
// the call are then visible to that workgroup's ld_agent loads.  The ticket is left at 0 for the next launch.
__device__ __forceinline__ bool last_arriver(int* ticket, int total) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's write-through stores have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == total - 1;
        if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_last != 0;
}

// The eight per-thread sums of a reduction kernel (values 0-3: first slot row, 4-7: second) -> this workgroup's slot
// [2][C]: lanes that own the same column group meet by wave shuffles (q = C/4 divides 64), the 16 waves through LDS.
template <bool AGENT = true>
__device__ __forceinline__ void publish_slot(float (&v)[8], float* slot, int C, float (*red)[16][8]) {
    const int q = C >> 2;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        for (int o = 32; o >= q; o >>= 1) v[j] += __shfl_down(v[j], o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < q) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][lane][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < q * 8) {
        const int grp = threadIdx.x >> 3, val = threadIdx.x & 7;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < kRedThreads / 64; ++w) s += red[w][grp][val];
        // (AGENT: read by another workgroup of the SAME launch -> write-through; otherwise by the next launch -> plain)
        if (AGENT) st_agent(slot + (val >> 2) * C + grp * 4 + (val & 3), s);
        else slot[(val >> 2) * C + grp * 4 + (val & 3)] = s;
    }
}

// Last workgroup: tot[g * 2C + col] (fp64, LDS) = the sum over the nblk slots of group g, for every group and column;
// slots are walked in index order by a fixed number of lanes per column: deterministic.
constexpr int kMaxPairs = 2048;            // groups * 2C the finishing workgroup holds (16 views x 64 channels)
template <bool AGENT = true>
__device__ __forceinline__ void sum_slots(const float* __restrict__ partial, int nblk, int groups, int C, double* tot,
                                          double* red) {
    const int ncol = 2 * C, pairs = groups * ncol;
    for (int base = 0; base < pairs; base += kRedThreads) {
        const int np = pairs - base < kRedThreads ? pairs - base : kRedThreads;
        int lanes = 1;
        while (lanes * 2 * np <= kRedThreads) lanes *= 2;
        const int pair = threadIdx.x % np, lane = threadIdx.x / np;
        double s = 0.0;
        if (lane < lanes) {
            const int g = (base + pair) / ncol, col = (base + pair) - g * ncol;
            const float* pg = partial + (long)g * nblk * ncol + col;
            // (sixteen independent loads in flight per round: the slots were written through to memory, a load is ~1-2 us)
            for (int n = lane; n < nblk; n += lanes * 16) {
                float t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int m = n + u * lanes;
                    t[u] = m < nblk ? (AGENT ? ld_agent(pg + (long)m * ncol) : pg[(long)m * ncol]) : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) s += (double)t[u];
            }
        }
        __syncthreads();
        red[threadIdx.x] = s;
        __syncthreads();
        // (the column's first thread adds its <= 64 lane sums in order: two barriers instead of a tree's seven)
        if (lane == 0) {
            double t = red[pair];
            for (int l = 1; l < lanes; ++l) t += red[l * np + pair];
            tot[base + pair] = t;
        }
    }
    __syncthreads();
}

// blockIdx.y = statistics group (the reference normalises every view's batch separately): group g owns rows
// [g*rows, (g+1)*rows) of x and row g of the per-channel parameter arrays.
__global__ void __launch_bounds__(256) bn_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ skip,
                                                          float* __restrict__ y, long n4, int C, int relu) {
    x += (long)blockIdx.y * n4 * 4; y += (long)blockIdx.y * n4 * 4;
    if (skip) skip += (long)blockIdx.y * n4 * 4;
    scale += blockIdx.y * C; shift += blockIdx.y * C;
    const int q = C >> 2;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 sc = ld4(scale + cg), sh = ld4(shift + cg);
    for (; i < n4; i += stride) {
        const f32x4 v = ld4(x + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = bn_act(v[j], sc[j], sh[j]);
            o[j] = relu ? fmaxf(t, 0.0f) : t;
        }
        if (skip) o += ld4(skip + i * 4);        // U-Net skip connection, added after the activation (mvs4net_utils.py:893-895)
        st4(y + i * 4, o);
    }
}

// Batch statistics, one launch: per (group, channel) sums of (x - p) and (x - p)^2 with the pivot p = the group's
// first row (keeps the E[d^2] - E[d]^2 subtraction well conditioned whatever the channel's mean is); the last workgroup
// then writes out [5][groups][C] = mean, biased variance, rstd, scale, shift and applies the running-average updates of the
// groups one after the other (what `groups` sequential module calls would do: momentum, unbiased variance) and
// num_batches_tracked += groups.  blockIdx.y = statistics group.
struct BnStatsArgs {
    const float* x; const float* weight; const float* bias;
    float* running_mean; float* running_var; long* num_batches_tracked; float* partial; float* out; int* ticket;
    long rows, n4; int C, groups; float eps, momentum;
};

__global__ void __launch_bounds__(kRedThreads) bn_stats_kernel(BnStatsArgs a) {
    __shared__ float red[kRedThreads / 64][16][8];
    __shared__ double dred[kRedThreads];
    __shared__ double tot[kMaxPairs];
    const int C = a.C, q = C >> 2, nblk = gridDim.x;
    const float* x = a.x + (long)blockIdx.y * a.n4 * 4;
    const long stride = (long)nblk * kRedThreads;
    long i = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 pv = ld4(x + cg);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; i < a.n4; i += stride) {
        const f32x4 t = ld4(x + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = t[j] - pv[j];
            v[j] += d;
            v[4 + j] = fmaf(d, d, v[4 + j]);
        }
    }
    publish_slot(v, a.partial + ((long)blockIdx.y * nblk + blockIdx.x) * 2 * C, C, red);
    if (!last_arriver(a.ticket, nblk * a.groups)) return;
    sum_slots(a.partial, nblk, a.groups, C, tot, dred);
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        const bool running = a.running_mean != nullptr;
        const float unbias = (float)a.rows / (float)(a.rows > 1 ? a.rows - 1 : 1);
        float rm = 0.0f, rv = 0.0f;
        if (running) { rm = a.running_mean[c]; rv = a.running_var[c]; }
        for (int g = 0; g < a.groups; ++g) {
            const double m1 = tot[g * 2 * C + c] / (double)a.rows, m2 = tot[g * 2 * C + C + c] / (double)a.rows;
            const float mean = a.x[(long)g * a.rows * C + c] + (float)m1;
            float var = (float)(m2 - m1 * m1);
            var = var > 0.0f ? var : 0.0f;
            const float rstd = 1.0f / sqrtf(var + a.eps);
            const float scale = a.weight[c] * rstd;
            float* o = a.out + (long)g * C + c;
            const long gs = (long)a.groups * C;
            o[0] = mean; o[gs] = var; o[2 * gs] = rstd; o[3 * gs] = scale; o[4 * gs] = a.bias[c] - mean * scale;
            rm = (1.0f - a.momentum) * rm + a.momentum * mean;
            rv = (1.0f - a.momentum) * rv + a.momentum * (var * unbias);
        }
        if (running) { a.running_mean[c] = rm; a.running_var[c] = rv; }
    }
    if (threadIdx.x == 0 && a.num_batches_tracked) *a.num_batches_tracked += a.groups;
}

// Backward sums, one launch: the last workgroup writes sums [groups][2][C] for the apply kernel and the parameter
// gradients summed over the groups, dbeta [C] = sum_g sum g_, dgamma [C] = sum_g sum g_*xh.
struct BnBwdReduceArgs {
    const float* x; const float* gy; const float* scale; const float* shift; const float* mean; const float* rstd;
    float* partial; float* sums; float* dgamma; float* dbeta; int* ticket;
    long n4; int C, relu, groups;
};

__global__ void __launch_bounds__(kRedThreads) bn_relu_bwd_reduce_kernel(BnBwdReduceArgs a) {
    __shared__ float red[kRedThreads / 64][16][8];
    __shared__ double dred[kRedThreads];
    __shared__ double tot[kMaxPairs];
    const int C = a.C, q = C >> 2, nblk = gridDim.x, relu = a.relu;
    const float* x = a.x + (long)blockIdx.y * a.n4 * 4;
    const float* gy = a.gy + (long)blockIdx.y * a.n4 * 4;
    const float *scale = a.scale + blockIdx.y * C, *shift = a.shift + blockIdx.y * C, *mean = a.mean + blockIdx.y * C,
                *rstd = a.rstd + blockIdx.y * C;
    const long stride = (long)nblk * kRedThreads;
    long i = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 sc = ld4(scale + cg), sh = ld4(shift + cg), mu = ld4(mean + cg), rs = ld4(rstd + cg);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; i < a.n4; i += stride) {
        const f32x4 t = ld4(x + i * 4), g4 = ld4(gy + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = g4[j] * relu_mask(relu, bn_act(t[j], sc[j], sh[j]));
            v[j] += g;
            v[4 + j] = fmaf(g, (t[j] - mu[j]) * rs[j], v[4 + j]);
        }
    }
    publish_slot(v, a.partial + ((long)blockIdx.y * nblk + blockIdx.x) * 2 * C, C, red);
    if (!last_arriver(a.ticket, nblk * a.groups)) return;
    sum_slots(a.partial, nblk, a.groups, C, tot, dred);
    for (int k = threadIdx.x; k < a.groups * 2 * C; k += kRedThreads) a.sums[k] = (float)tot[k];
    if (threadIdx.x < 2 * C) {
        double total = 0.0;
        for (int g = 0; g < a.groups; ++g) total += tot[g * 2 * C + threadIdx.x];
        if (threadIdx.x < C) a.dbeta[threadIdx.x] = (float)total;
        else a.dgamma[threadIdx.x - C] = (float)total;
    }
}

// Column sums of a channels-last tensor [rows, C] -> out [C] (the bias gradient of a convolution, sum of gy over every
// voxel): the reductions' machinery with one statistics group, one launch, deterministic.
struct ColSumArgs {
    const float* x; float* partial; float* out; int* ticket;
    long n4; int C;
};

__global__ void __launch_bounds__(kRedThreads) col_sum_kernel(ColSumArgs a) {
    __shared__ float red[kRedThreads / 64][16][8];
    __shared__ double dred[kRedThreads];
    __shared__ double tot[2 * 64];
    const int C = a.C, q = C >> 2, nblk = gridDim.x;
    const long stride = (long)nblk * kRedThreads;
    long i = (long)blockIdx.x * kRedThreads + threadIdx.x;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // two rows in flight per thread (second accumulator row: slot row 1, added to row 0 by the finishing workgroup)
    for (; i + stride < a.n4; i += 2 * stride) {
        const f32x4 t0 = ld4(a.x + i * 4), t1 = ld4(a.x + (i + stride) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += t0[j]; v[4 + j] += t1[j]; }
    }
    if (i < a.n4) {
        const f32x4 t0 = ld4(a.x + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += t0[j];
    }
    (void)q;
    publish_slot(v, a.partial + (long)blockIdx.x * 2 * C, C, red);
    if (!last_arriver(a.ticket, nblk)) return;
    sum_slots(a.partial, nblk, 1, C, tot, dred);
    if (threadIdx.x < C) a.out[threadIdx.x] = (float)(tot[threadIdx.x] + tot[C + threadIdx.x]);
}

__global__ void __launch_bounds__(256) bn_relu_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ sums, float* __restrict__ dx,
                                                                long n4, int C, int relu, float inv_n) {
    x += (long)blockIdx.y * n4 * 4; gy += (long)blockIdx.y * n4 * 4; dx += (long)blockIdx.y * n4 * 4;
    scale += blockIdx.y * C; shift += blockIdx.y * C; mean += blockIdx.y * C; rstd += blockIdx.y * C;
    sums += blockIdx.y * 2 * C;
    const int q = C >> 2;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 sc = ld4(scale + cg), sh = ld4(shift + cg), mu = ld4(mean + cg), rs = ld4(rstd + cg);
    const f32x4 s0 = ld4(sums + cg), s1 = ld4(sums + C + cg);
    for (; i < n4; i += stride) {
        const f32x4 v = ld4(x + i * 4), g4 = ld4(gy + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = g4[j] * relu_mask(relu, bn_act(v[j], sc[j], sh[j]));
            const float xh = (v[j] - mu[j]) * rs[j];
            o[j] = sc[j] * (g - s0[j] * inv_n - xh * s1[j] * inv_n);
        }
        st4(dx + i * 4, o);
    }
}

// ---- the training passes without a serial tail: slots in one launch, their sum in the NEXT launch's prologue --------------
// The one-launch reductions above end in a tail only the last workgroup runs -- store drain, ticket, one or two rounds of
// memory-latency loads of the written-through slots, an LDS tree, the finalize: 8-10 us per launch, 108 launches per step,
// more than the streaming of most layers.  Here the reduction kernels just write their slots (plain stores: the next
// launch finds them in L2) and the APPLY kernels sum them in their prologue -- every workgroup its own group's, in the same
// fixed order (deterministic, every workgroup gets the same bits), ~2 us, all workgroups at once.  One extra workgroup per
// launch does what needs all groups in order: the running averages (forward), the parameter gradients (backward).
struct BnSlotsArgs {
    const float* x; const float* gy; const float* pack; float* partial;
    long n4; int C, groups, relu;
};

// forward statistics slots: sums of (x - pivot), (x - pivot)^2 (pivot = the group's first row)
__global__ void __launch_bounds__(kRedThreads) bn_stats_slots_kernel(BnSlotsArgs a) {
    __shared__ float red[kRedThreads / 64][16][8];
    const int C = a.C, q = C >> 2, nblk = gridDim.x;
    const float* x = a.x + (long)blockIdx.y * a.n4 * 4;
    const long stride = (long)nblk * kRedThreads;
    long i = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 pv = ld4(x + cg);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; i < a.n4; i += stride) {
        const f32x4 t = ld4(x + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = t[j] - pv[j];
            v[j] += d;
            v[4 + j] = fmaf(d, d, v[4 + j]);
        }
    }
    publish_slot<false>(v, a.partial + ((long)blockIdx.y * nblk + blockIdx.x) * 2 * C, C, red);
}

// backward slots: sums of g = gy * (y > 0) and g * xh
__global__ void __launch_bounds__(kRedThreads) bn_bwd_slots_kernel(BnSlotsArgs a) {
    __shared__ float red[kRedThreads / 64][16][8];
    const int C = a.C, q = C >> 2, nblk = gridDim.x, relu = a.relu, g = blockIdx.y;
    const long gs = (long)a.groups * C;
    const float* x = a.x + (long)g * a.n4 * 4;
    const float* gy = a.gy + (long)g * a.n4 * 4;
    const long stride = (long)nblk * kRedThreads;
    long i = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 mu = ld4(a.pack + (long)g * C + cg), rs = ld4(a.pack + 2 * gs + (long)g * C + cg);
    const f32x4 sc = ld4(a.pack + 3 * gs + (long)g * C + cg), sh = ld4(a.pack + 4 * gs + (long)g * C + cg);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; i < a.n4; i += stride) {
        const f32x4 t = ld4(x + i * 4), g4 = ld4(gy + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = g4[j] * relu_mask(relu, bn_act(t[j], sc[j], sh[j]));
            v[j] += gg;
            v[4 + j] = fmaf(gg, (t[j] - mu[j]) * rs[j], v[4 + j]);
        }
    }
    publish_slot<false>(v, a.partial + ((long)g * nblk + blockIdx.x) * 2 * C, C, red);
}

struct BnTrainFwdArgs {
    const float* x; const float* partial; const float* weight; const float* bias; const float* skip;
    float* running_mean; float* running_var; long* num_batches_tracked; float* y; float* out;
    long rows, n4; int C, groups, relu, nslots; float eps, momentum;
};

// mean / biased variance of (group gg, channel c) from the summed slots
__device__ __forceinline__ void bn_moments(const double* tot, int C, int c, long rows, float pivot, float& mean, float& var) {
    const double m1 = tot[c] / (double)rows, m2 = tot[C + c] / (double)rows;
    mean = pivot + (float)m1;
    var = (float)(m2 - m1 * m1);
    var = var > 0.0f ? var : 0.0f;
}

// grid (apply blocks + 1, groups): block (last, 0) applies the running-average updates of all groups in order (what `groups`
// sequential module calls would do) and counts the batches; the others finish THEIR group's statistics and stream.
__global__ void __launch_bounds__(kRedThreads) bn_train_fwd_kernel(BnTrainFwdArgs a) {
    __shared__ double dred[kRedThreads];
    __shared__ double tot[kMaxPairs];
    __shared__ float scl[64], shl[64];
    const int C = a.C, g = blockIdx.y, nblk = gridDim.x - 1;
    if ((int)blockIdx.x == nblk) {
        if (g != 0 || (!a.running_mean && !a.num_batches_tracked)) return;
        sum_slots<false>(a.partial, a.nslots, a.groups, C, tot, dred);
        if (threadIdx.x < C && a.running_mean) {
            const int c = threadIdx.x;
            const float unbias = (float)a.rows / (float)(a.rows > 1 ? a.rows - 1 : 1);
            float rm = a.running_mean[c], rv = a.running_var[c];
            for (int gg = 0; gg < a.groups; ++gg) {
                float mean, var;
                bn_moments(tot + gg * 2 * C, C, c, a.rows, a.x[(long)gg * a.rows * C + c], mean, var);
                rm = (1.0f - a.momentum) * rm + a.momentum * mean;
                rv = (1.0f - a.momentum) * rv + a.momentum * (var * unbias);
            }
            a.running_mean[c] = rm; a.running_var[c] = rv;
        }
        if (threadIdx.x == 0 && a.num_batches_tracked) *a.num_batches_tracked += a.groups;
        return;
    }
    // (the first streaming loads are issued before the prologue: for most layers that is all a thread reads)
    const long ifirst = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const f32x4 vfirst = ifirst < a.n4 ? ld4(a.x + (long)g * a.n4 * 4 + ifirst * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    sum_slots<false>(a.partial + (long)g * a.nslots * 2 * C, a.nslots, 1, C, tot, dred);
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        float mean, var;
        bn_moments(tot, C, c, a.rows, a.x[(long)g * a.rows * C + c], mean, var);
        const float rstd = 1.0f / sqrtf(var + a.eps);
        const float scale = a.weight[c] * rstd, shift = a.bias[c] - mean * scale;
        scl[c] = scale; shl[c] = shift;
        if (blockIdx.x == 0) {
            float* o = a.out + (long)g * C + c;
            const long gs = (long)a.groups * C;
            o[0] = mean; o[gs] = var; o[2 * gs] = rstd; o[3 * gs] = scale; o[4 * gs] = shift;
        }
    }
    __syncthreads();
    const int q = C >> 2;
    const float* x = a.x + (long)g * a.n4 * 4;
    float* y = a.y + (long)g * a.n4 * 4;
    const float* skip = a.skip ? a.skip + (long)g * a.n4 * 4 : nullptr;
    const long stride = (long)nblk * kRedThreads;
    long i = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 sc = {scl[cg], scl[cg + 1], scl[cg + 2], scl[cg + 3]}, sh = {shl[cg], shl[cg + 1], shl[cg + 2], shl[cg + 3]};
    for (; i < a.n4; i += stride) {
        const f32x4 v = i == ifirst ? vfirst : ld4(x + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = bn_act(v[j], sc[j], sh[j]);
            o[j] = a.relu ? fmaxf(t, 0.0f) : t;
        }
        if (skip) o += ld4(skip + i * 4);
        st4(y + i * 4, o);
    }
}

struct BnTrainBwdArgs {
    const float* x; const float* gy; const float* pack; const float* partial; float* dgamma; float* dbeta; float* dx;
    long rows, n4; int C, groups, relu, nslots;
};

// grid (apply blocks + 1, groups): block (last, 0) writes dgamma / dbeta (sums over the groups); the others sum THEIR group's
// slots and stream dx = scale * (g - sum_g / n - xh * sum_gxh / n).
__global__ void __launch_bounds__(kRedThreads) bn_train_bwd_kernel(BnTrainBwdArgs a) {
    __shared__ double dred[kRedThreads];
    __shared__ double tot[kMaxPairs];
    __shared__ float s0l[64], s1l[64];
    const int C = a.C, g = blockIdx.y, nblk = gridDim.x - 1, relu = a.relu;
    if ((int)blockIdx.x == nblk) {
        if (g != 0) return;
        sum_slots<false>(a.partial, a.nslots, a.groups, C, tot, dred);
        if (threadIdx.x < 2 * C) {
            double t2 = 0.0;
            for (int gg = 0; gg < a.groups; ++gg) t2 += tot[gg * 2 * C + threadIdx.x];
            if (threadIdx.x < C) a.dbeta[threadIdx.x] = (float)t2;
            else a.dgamma[threadIdx.x - C] = (float)t2;
        }
        return;
    }
    const long ifirst = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const bool hasfirst = ifirst < a.n4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 vfirst = hasfirst ? ld4(a.x + (long)g * a.n4 * 4 + ifirst * 4) : zero4;
    const f32x4 gfirst = hasfirst ? ld4(a.gy + (long)g * a.n4 * 4 + ifirst * 4) : zero4;
    sum_slots<false>(a.partial + (long)g * a.nslots * 2 * C, a.nslots, 1, C, tot, dred);
    if (threadIdx.x < C) { s0l[threadIdx.x] = (float)tot[threadIdx.x]; s1l[threadIdx.x] = (float)tot[C + threadIdx.x]; }
    __syncthreads();
    const int q = C >> 2;
    const long gs = (long)a.groups * C;
    const float* x = a.x + (long)g * a.n4 * 4;
    const float* gy = a.gy + (long)g * a.n4 * 4;
    float* dx = a.dx + (long)g * a.n4 * 4;
    const long stride = (long)nblk * kRedThreads;
    long i = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i % q) * 4;
    const f32x4 mu = ld4(a.pack + (long)g * C + cg), rs = ld4(a.pack + 2 * gs + (long)g * C + cg);
    const f32x4 sc = ld4(a.pack + 3 * gs + (long)g * C + cg), sh = ld4(a.pack + 4 * gs + (long)g * C + cg);
    const f32x4 s0 = {s0l[cg], s0l[cg + 1], s0l[cg + 2], s0l[cg + 3]}, s1 = {s1l[cg], s1l[cg + 1], s1l[cg + 2], s1l[cg + 3]};
    const float inv_n = 1.0f / (float)a.rows;
    for (; i < a.n4; i += stride) {
        const f32x4 v = i == ifirst ? vfirst : ld4(x + i * 4), g4 = i == ifirst ? gfirst : ld4(gy + i * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = g4[j] * relu_mask(relu, bn_act(v[j], sc[j], sh[j]));
            const float xh = (v[j] - mu[j]) * rs[j];
            o[j] = sc[j] * (gg - s0[j] * inv_n - xh * s1[j] * inv_n);
        }
        st4(dx + i * 4, o);
    }
}

// ---- small tensors: statistics + apply in ONE launch (forward), reduce + apply in ONE launch (backward) -------------------
// Most BatchNorm layers of the step are small (coarse cascade stages, deep U-Net levels): two dependent launches each way
// cost more in latency than in bytes.  Here a thread keeps its R float4 of x (and gy) in registers across the reduction: the
// workgroups publish their slots, the last arriver finishes the statistics and publishes them (write-through) and raises a
// flag the others spin on (one lane per workgroup, s_sleep) -- a grid barrier, safe because the grid is at most
// kFusedMaxWG workgroups of 1024 threads (every one resident on its own CU) -- then everybody applies from registers.
// One read of x instead of two (three of x / gy instead of four in the backward).  sync = {arrivals, flag, departures}, all
// zero before the launch; the last workgroup to LEAVE zeroes them again.
constexpr int kFusedMaxWG = 128;

__device__ __forceinline__ int ld_agent_i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_i(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// arrive; true (in every thread) for the last workgroup, which must call grid_release() after publishing its results
__device__ __forceinline__ bool grid_arrive(int* sync, int total) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1;
    __syncthreads();
    return s_last != 0;
}
__device__ __forceinline__ void grid_release(int* sync) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's write-through result stores have completed
    __syncthreads();
    if (threadIdx.x == 0) st_agent_i(sync + 1, 1);
}
__device__ __forceinline__ void grid_wait(int* sync) {
    if (threadIdx.x == 0)
        while (ld_agent_i(sync + 1) == 0) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
}
__device__ __forceinline__ void grid_depart(int* sync, int total) {
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
        st_agent_i(sync, 0);
        st_agent_i(sync + 1, 0);
        st_agent_i(sync + 2, 0);
    }
}

struct BnFusedFwdArgs {
    const float* x; const float* skip; float* y; const float* weight; const float* bias;
    float* running_mean; float* running_var; long* num_batches_tracked; float* partial; float* out; int* sync;
    long rows, n4; int C, groups, relu; float eps, momentum;
};

template <int R>
__global__ void __launch_bounds__(kRedThreads) bn_fused_fwd_kernel(BnFusedFwdArgs a) {
    __shared__ float red[kRedThreads / 64][16][8];
    __shared__ double dred[kRedThreads];
    __shared__ double tot[kMaxPairs];
    const int C = a.C, q = C >> 2, nblk = gridDim.x, g = blockIdx.y, total = nblk * a.groups;
    const float* x = a.x + (long)g * a.n4 * 4;
    const long stride = (long)nblk * kRedThreads;
    const long i0 = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i0 % q) * 4;
    const f32x4 pv = ld4(x + cg);
    f32x4 xr[R];
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const long i = i0 + j * stride;
        xr[j] = i < a.n4 ? ld4(x + i * 4) : pv;               // (the pivot adds nothing to the sums)
    }
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = xr[j][k] - pv[k];
            v[k] += d;
            v[4 + k] = fmaf(d, d, v[4 + k]);
        }
    publish_slot(v, a.partial + ((long)g * nblk + blockIdx.x) * 2 * C, C, red);
    const long gs = (long)a.groups * C;
    if (grid_arrive(a.sync, total)) {
        sum_slots(a.partial, nblk, a.groups, C, tot, dred);
        if (threadIdx.x < C) {
            const int c = threadIdx.x;
            const bool running = a.running_mean != nullptr;
            const float unbias = (float)a.rows / (float)(a.rows > 1 ? a.rows - 1 : 1);
            float rm = 0.0f, rv = 0.0f;
            if (running) { rm = a.running_mean[c]; rv = a.running_var[c]; }
            for (int gg = 0; gg < a.groups; ++gg) {
                const double m1 = tot[gg * 2 * C + c] / (double)a.rows, m2 = tot[gg * 2 * C + C + c] / (double)a.rows;
                const float mean = a.x[(long)gg * a.rows * C + c] + (float)m1;
                float var = (float)(m2 - m1 * m1);
                var = var > 0.0f ? var : 0.0f;
                const float rstd = 1.0f / sqrtf(var + a.eps);
                const float scale = a.weight[c] * rstd;
                float* o = a.out + (long)gg * C + c;
                st_agent(o, mean); st_agent(o + gs, var); st_agent(o + 2 * gs, rstd); st_agent(o + 3 * gs, scale);
                st_agent(o + 4 * gs, a.bias[c] - mean * scale);
                rm = (1.0f - a.momentum) * rm + a.momentum * mean;
                rv = (1.0f - a.momentum) * rv + a.momentum * (var * unbias);
            }
            if (running) { a.running_mean[c] = rm; a.running_var[c] = rv; }
        }
        if (threadIdx.x == 0 && a.num_batches_tracked) *a.num_batches_tracked += a.groups;
        grid_release(a.sync);
    } else {
        grid_wait(a.sync);
    }
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = ld_agent(a.out + 3 * gs + (long)g * C + cg + k);
        sh[k] = ld_agent(a.out + 4 * gs + (long)g * C + cg + k);
    }
    float* y = a.y + (long)g * a.n4 * 4;
    const float* skip = a.skip ? a.skip + (long)g * a.n4 * 4 : nullptr;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const long i = i0 + j * stride;
        if (i < a.n4) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = bn_act(xr[j][k], sc[k], sh[k]);
                o[k] = a.relu ? fmaxf(t, 0.0f) : t;
            }
            if (skip) o += ld4(skip + i * 4);
            st4(y + i * 4, o);
        }
    }
    grid_depart(a.sync, total);
}

struct BnFusedBwdArgs {
    const float* x; const float* gy; const float* pack; float* partial; float* sums; float* dgamma; float* dbeta; float* dx;
    int* sync; long rows, n4; int C, groups, relu;
};

template <int R>
__global__ void __launch_bounds__(kRedThreads) bn_fused_bwd_kernel(BnFusedBwdArgs a) {
    __shared__ float red[kRedThreads / 64][16][8];
    __shared__ double dred[kRedThreads];
    __shared__ double tot[kMaxPairs];
    const int C = a.C, q = C >> 2, nblk = gridDim.x, g = blockIdx.y, total = nblk * a.groups, relu = a.relu;
    const long gs = (long)a.groups * C;
    const float* x = a.x + (long)g * a.n4 * 4;
    const float* gy = a.gy + (long)g * a.n4 * 4;
    const long stride = (long)nblk * kRedThreads;
    const long i0 = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const int cg = (int)(i0 % q) * 4;
    const f32x4 mu = ld4(a.pack + (long)g * C + cg), rs = ld4(a.pack + 2 * gs + (long)g * C + cg);
    const f32x4 sc = ld4(a.pack + 3 * gs + (long)g * C + cg), sh = ld4(a.pack + 4 * gs + (long)g * C + cg);
    f32x4 gr[R], xh[R];                                           // g = gy * mask, xh = (x - mean) * rstd
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const long i = i0 + j * stride;
        const bool ok = i < a.n4;
        const f32x4 t = ok ? ld4(x + i * 4) : mu, g4 = ok ? ld4(gy + i * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gr[j][k] = g4[k] * relu_mask(relu, bn_act(t[k], sc[k], sh[k]));
            xh[j][k] = (t[k] - mu[k]) * rs[k];
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] += gr[j][k];
            v[4 + k] = fmaf(gr[j][k], xh[j][k], v[4 + k]);
        }
    publish_slot(v, a.partial + ((long)g * nblk + blockIdx.x) * 2 * C, C, red);
    if (grid_arrive(a.sync, total)) {
        sum_slots(a.partial, nblk, a.groups, C, tot, dred);
        for (int k = threadIdx.x; k < a.groups * 2 * C; k += kRedThreads) st_agent(a.sums + k, (float)tot[k]);
        if (threadIdx.x < 2 * C) {
            double t2 = 0.0;
            for (int gg = 0; gg < a.groups; ++gg) t2 += tot[gg * 2 * C + threadIdx.x];
            if (threadIdx.x < C) a.dbeta[threadIdx.x] = (float)t2;
            else a.dgamma[threadIdx.x - C] = (float)t2;
        }
        grid_release(a.sync);
    } else {
        grid_wait(a.sync);
    }
    const float inv_n = 1.0f / (float)a.rows;
    float s0[4], s1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s0[k] = ld_agent(a.sums + (long)g * 2 * C + cg + k);
        s1[k] = ld_agent(a.sums + (long)g * 2 * C + C + cg + k);
    }
    float* dx = a.dx + (long)g * a.n4 * 4;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const long i = i0 + j * stride;
        if (i < a.n4) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = sc[k] * (gr[j][k] - s0[k] * inv_n - xh[j][k] * s1[k] * inv_n);
            st4(dx + i * 4, o);
        }
    }
    grid_depart(a.sync, total);
}

// float4 per thread (R) and workgroups per group for the fused forms, or R = 0 where the tensor does not fit the resident grid
int fused_plan(long rows, int C, int groups, int rmax, int& nblk) {
    const long n4 = rows * (C / 4);
    if (groups < 1 || groups > kFusedMaxWG || groups * 2 * C > kMaxPairs) return 0;
    const int cap = kFusedMaxWG / groups;
    for (int r = 2; r <= rmax; r *= 2) {
        const long n = (n4 + (long)kRedThreads * r - 1) / ((long)kRedThreads * r);
        if (n <= cap) { nblk = (int)(n < 1 ? 1 : n); return r; }
    }
    return 0;
}

int check(long rows, int C) {
    if (rows <= 0) return MVSTER_ERR_SHAPE;
    if (C < 4 || C > 64 || (C & (C - 1)) != 0) return MVSTER_ERR_UNSUPPORTED;
    return MVSTER_OK;
}

// Slots (= workgroups of 1024 threads) per group of `partial` for the two reductions: at most one workgroup per CU over
// all groups (16 waves per CU stream at full rate, and the last arriver adds <= 256 slots).
int slots_for(long rows, int C, int groups) {
    const long n4 = rows * (C / 4);
    // (at least four grid-stride rounds per workgroup: the last arriver's tail grows with the slots -- 17.8 against 8.4 us
    //  of streaming for a 10 MB, 64-channel tensor with 256 slots -- and a small tensor is latency-bound anyway)
    long n = (n4 + 4 * kRedThreads - 1) / (4 * kRedThreads);
    const long cap = groups >= 256 ? 1 : 256 / groups;
    if (n > cap) n = cap;
    return (int)(n < 1 ? 1 : n);
}

int train_apply_blocks(long rows, int C, int groups);

// Slots of the tail-less training passes: every apply workgroup sums all of its group's slots, 2C columns each, so wide
// tensors get fewer (<= 8 loads per thread of the 1024: 64 slots at 64 channels, 128 at 32, 256 below)
int train_slots_for(long rows, int C, int groups) {
    int n = slots_for(rows, C, groups);               // (at most one per CU: with 512 slots at 8 channels the apply kernels'
                                                      //  prologues cost more than the streaming gains, 2.41 against 2.32 ms)
    const int cap = 4096 / C;
    return n > cap ? cap : n;
}

// Streaming workgroups (1024 threads) of the tail-less apply kernels: two per CU for the large tensors
int train_apply_blocks(long rows, int C, int groups) {
    const long n4 = rows * (C / 4);
    long n = (n4 + 4 * kRedThreads - 1) / (4 * kRedThreads);
    const long cap = groups >= 512 ? 1 : 512 / groups;
    if (n > cap) n = cap;
    return (int)(n < 1 ? 1 : n);
}

int blocks_for(long n4) {
    const long want = (n4 + 255) / 256;
    return (int)(want < 2048 ? want : 2048);
}

}  // namespace

// rows = rows PER GROUP; x, y [groups*rows, C]; scale, shift (mean, rstd) [groups, C]
extern "C" int mvster_bn_relu_fwd(const float* x, const float* scale, const float* shift, const float* skip, float* y,
                                  long rows, int C, int relu, int groups, void* stream) {
    if (!x || !scale || !shift || !y) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(blocks_for(n4), groups), dim3(256), 0, (hipStream_t)stream, x, scale, shift, skip,
                       y, n4, C, relu);
    return mv_check_launch();
}

// partial [groups][mvster_bn_slots(rows, C, groups)][2][C] floats (scratch of stats and bwd_reduce)
extern "C" int mvster_bn_slots(long rows, int C, int groups) {
    if (check(rows, C) || groups < 1 || groups > 65535) return 0;
    return slots_for(rows, C, groups);
}

// out [5][groups][C] = mean, biased var, rstd, scale = gamma*rstd, shift = beta - mean*scale; running_mean / running_var
// (optional, [C]) receive the `groups` exponential-average updates in group order (unbiased variance, torch semantics)
// and num_batches_tracked (optional, int64 on the device) += groups.  ticket: one int on the device, 0 before the call
// and 0 again after it (the launch's arrival counter).  One launch; groups * C <= 1024.
extern "C" int mvster_bn_stats(const float* x, const float* weight, const float* bias, float* running_mean,
                               float* running_var, long* num_batches_tracked, float* partial, float* out, int* ticket,
                               long rows, int C, int groups, float eps, float momentum, void* stream) {
    if (!x || !weight || !bias || !partial || !out || !ticket) return MVSTER_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    if (groups * 2 * C > kMaxPairs) return MVSTER_ERR_UNSUPPORTED;
    const int nblk = slots_for(rows, C, groups);
    BnStatsArgs a{x, weight, bias, running_mean, running_var, num_batches_tracked, partial, out, ticket,
                  rows, rows * (C / 4), C, groups, eps, momentum};
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk, groups), dim3(kRedThreads), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

// sums [groups][2][C] (for bwd_apply), dgamma [C], dbeta [C]; partial and ticket as for mvster_bn_stats.  One launch.
extern "C" int mvster_bn_relu_bwd_reduce(const float* x, const float* gy, const float* scale, const float* shift,
                                         const float* mean, const float* rstd, float* partial, float* sums, float* dgamma,
                                         float* dbeta, int* ticket, long rows, int C, int relu, int groups, void* stream) {
    if (!x || !gy || !scale || !shift || !mean || !rstd || !partial || !sums || !dgamma || !dbeta || !ticket)
        return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    if (groups * 2 * C > kMaxPairs) return MVSTER_ERR_UNSUPPORTED;
    const int nblk = slots_for(rows, C, groups);
    BnBwdReduceArgs a{x, gy, scale, shift, mean, rstd, partial, sums, dgamma, dbeta, ticket, rows * (C / 4), C, relu, groups};
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(nblk, groups), dim3(kRedThreads), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

// sums [groups][2][C] = (sum g, sum g*xh) over the group's rows; dx [groups*rows, C].  frozen = 1: mean / rstd are
// constants (running statistics of a BatchNorm in eval mode), so dx = g * scale and the sums are not applied.
extern "C" int mvster_bn_relu_bwd_apply(const float* x, const float* gy, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, const float* sums, float* dx, long rows, int C,
                                        int relu, int groups, int frozen, void* stream) {
    if (!x || !gy || !scale || !shift || !mean || !rstd || !sums || !dx) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535) return MVSTER_ERR_SHAPE;
    const long n4 = rows * (C / 4);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(blocks_for(n4), groups), dim3(256), 0, (hipStream_t)stream, x, gy, scale,
                       shift, mean, rstd, sums, dx, n4, C, relu, frozen ? 0.0f : 1.0f / (float)rows);
    return mv_check_launch();
}

// out [C] = column sums of x [rows, C] (C in {4, 8, 16, 32, 64}); partial: mvster_bn_slots(rows, C, 1) * 2 * C floats of
// scratch, ticket as for mvster_bn_stats.  The bias gradient of the reference's convolutions with bias (FPN laterals,
// monocular heads: models/mvs4net_utils.py:485-487, :846-848).  One launch.
extern "C" int mvster_col_sum(const float* x, float* partial, float* out, int* ticket, long rows, int C, void* stream) {
    if (!x || !partial || !out || !ticket) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    ColSumArgs a{x, partial, out, ticket, rows * (C / 4), C};
    hipLaunchKernelGGL(col_sum_kernel, dim3(slots_for(rows, C, 1)), dim3(kRedThreads), 0, (hipStream_t)stream, a);
    return mv_check_launch();
}

// Whether mvster_bn_fwd_fused (backward = 0) / mvster_bn_bwd_fused (backward = 1) take a tensor of `groups` x `rows` x C: the
// grid must be resident as a whole (<= 128 workgroups of 1024 threads, <= 8 / 4 float4 of x (and gy) per thread: 16 / 8 MB;
// twice that spills registers at 1024 threads per workgroup).
extern "C" int mvster_bn_fused_ok(long rows, int C, int groups, int backward) {
    if (check(rows, C)) return 0;
    int nblk = 0;
    return fused_plan(rows, C, groups, backward ? 4 : 8, nblk) ? 1 : 0;
}

// Training-mode BatchNorm (+ ReLU, + skip) of a SMALL tensor in one launch: what mvster_bn_stats + mvster_bn_relu_fwd do in
// two (same statistics pack `out` [5][groups][C], same running-average updates).  partial: groups * 128 * 2 * C floats of
// scratch; sync: 3 ints on the device, zero before the call and zero again after it.  MVSTER_ERR_UNSUPPORTED where
// mvster_bn_fused_ok says no.
extern "C" int mvster_bn_fwd_fused(const float* x, const float* skip, float* y, const float* weight, const float* bias,
                                   float* running_mean, float* running_var, long* num_batches_tracked, float* partial, float* out,
                                   int* sync, long rows, int C, int relu, int groups, float eps, float momentum, void* stream) {
    if (!x || !y || !weight || !bias || !partial || !out || !sync) return MVSTER_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    int nblk = 0;
    const int r = fused_plan(rows, C, groups, 8, nblk);
    if (!r) return MVSTER_ERR_UNSUPPORTED;
    BnFusedFwdArgs a{x, skip, y, weight, bias, running_mean, running_var, num_batches_tracked, partial, out, sync,
                     rows, rows * (C / 4), C, groups, relu, eps, momentum};
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblk, groups), block(kRedThreads);
    if (r == 2) hipLaunchKernelGGL(bn_fused_fwd_kernel<2>, grid, block, 0, s, a);
    else if (r == 4) hipLaunchKernelGGL(bn_fused_fwd_kernel<4>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(bn_fused_fwd_kernel<8>, grid, block, 0, s, a);
    return mv_check_launch();
}

// Its backward in one launch (mvster_bn_relu_bwd_reduce + mvster_bn_relu_bwd_apply): pack = the forward's `out`; sums
// [groups][2][C] scratch; dgamma, dbeta [C]; dx like x.
extern "C" int mvster_bn_bwd_fused(const float* x, const float* gy, const float* pack, float* partial, float* sums, float* dgamma,
                                   float* dbeta, float* dx, int* sync, long rows, int C, int relu, int groups, void* stream) {
    if (!x || !gy || !pack || !partial || !sums || !dgamma || !dbeta || !dx || !sync) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    int nblk = 0;
    const int r = fused_plan(rows, C, groups, 4, nblk);
    if (!r) return MVSTER_ERR_UNSUPPORTED;
    BnFusedBwdArgs a{x, gy, pack, partial, sums, dgamma, dbeta, dx, sync, rows, rows * (C / 4), C, groups, relu};
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(nblk, groups), block(kRedThreads);
    if (r == 2) hipLaunchKernelGGL(bn_fused_bwd_kernel<2>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(bn_fused_bwd_kernel<4>, grid, block, 0, s, a);
    return mv_check_launch();
}

// slots per group of `partial` for mvster_bn_train_fwd / _bwd (0: unsupported channel count)
extern "C" int mvster_bn_train_slots(long rows, int C, int groups) {
    if (check(rows, C) || groups < 1 || groups > 65535) return 0;
    return train_slots_for(rows, C, groups);
}

// Training-mode BatchNorm (+ ReLU, + skip) as TWO launches without a serial tail: the statistics slots, then the apply kernel
// whose workgroups sum the slots in their prologue (see the kernels).  Same results as mvster_bn_stats + mvster_bn_relu_fwd
// (same slot sums, same fp64 finish, same running-average updates); out = the statistics pack [5][groups][C].  partial:
// groups * mvster_bn_train_slots(rows, C, groups) * 2 * C floats.
extern "C" int mvster_bn_train_fwd(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                                   long* num_batches_tracked, const float* skip, float* partial, float* y, float* out, long rows,
                                   int C, int relu, int groups, float eps, float momentum, void* stream) {
    if (!x || !weight || !bias || !partial || !y || !out) return MVSTER_ERR_NULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535 || groups * 2 * C > kMaxPairs) return groups < 1 ? MVSTER_ERR_SHAPE : MVSTER_ERR_UNSUPPORTED;
    const long n4 = rows * (C / 4);
    const int nslots = train_slots_for(rows, C, groups), napply = train_apply_blocks(rows, C, groups);
    hipStream_t s = (hipStream_t)stream;
    BnSlotsArgs sa{x, nullptr, nullptr, partial, n4, C, groups, relu};
    hipLaunchKernelGGL(bn_stats_slots_kernel, dim3(nslots, groups), dim3(kRedThreads), 0, s, sa);
    BnTrainFwdArgs a{x, partial, weight, bias, skip, running_mean, running_var, num_batches_tracked, y, out,
                     rows, n4, C, groups, relu, nslots, eps, momentum};
    hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(napply + 1, groups), dim3(kRedThreads), 0, s, a);
    return mv_check_launch();
}

// Its backward, two launches: the slots of sum g / sum g*xh, then dx with the sums formed in the prologue; pack = the
// forward's `out`; dgamma, dbeta [C] (summed over the groups).
extern "C" int mvster_bn_train_bwd(const float* x, const float* gy, const float* pack, float* partial, float* dgamma,
                                   float* dbeta, float* dx, long rows, int C, int relu, int groups, void* stream) {
    if (!x || !gy || !pack || !partial || !dgamma || !dbeta || !dx) return MVSTER_ERR_NULL;
    if (int rc = check(rows, C)) return rc;
    if (groups < 1 || groups > 65535 || groups * 2 * C > kMaxPairs) return groups < 1 ? MVSTER_ERR_SHAPE : MVSTER_ERR_UNSUPPORTED;
    const long n4 = rows * (C / 4);
    const int nslots = train_slots_for(rows, C, groups), napply = train_apply_blocks(rows, C, groups);
    hipStream_t s = (hipStream_t)stream;
    BnSlotsArgs sa{x, gy, pack, partial, n4, C, groups, relu};
    hipLaunchKernelGGL(bn_bwd_slots_kernel, dim3(nslots, groups), dim3(kRedThreads), 0, s, sa);
    BnTrainBwdArgs a{x, gy, pack, partial, dgamma, dbeta, dx, rows, n4, C, groups, relu, nslots};
    hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(napply + 1, groups), dim3(kRedThreads), 0, s, a);
    return mv_check_launch();
}
