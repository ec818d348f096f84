// Fused BatchNorm2d(+ReLU) for channels-last feature maps on gfx950.  C ABI + reference lines: include/liso_bn.h.
//
// HBM-bound elementwise/reduction passes.  Every thread moves 16 B per access (4 fp32 / 8 bf16 channels of one
// pixel); a 256-thread block covers 256/(C/V) pixel rows per step, so a wave-instruction reads whole 128/256-B
// channel rows back to back (fully coalesced).  Statistics: every block accumulates sums shifted by its own first row
// (so they stay well conditioned), emits (mean_b, M2_b), and the grid-level merge is the two-pass form over the block
// means in fp64 and a fixed order: the variance never suffers the E[x^2]-E[x]^2 cancellation and results are bitwise
// reproducible.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_bn.h"
#include "../../include/liso_iou3d.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerBlock = 128;
constexpr int kMaxBlocks = 4096;

template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int V = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec<__hip_bfloat16> {
    static constexpr int V = 8;
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[8]) {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]), hi = __float2bfloat16(v[2 * i + 1]);
            w[i] = (unsigned)(*reinterpret_cast<const unsigned short*>(&lo)) |
                   ((unsigned)(*reinterpret_cast<const unsigned short*>(&hi)) << 16);
        }
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

struct Geom {
    int cg;        // column groups = C / V
    int rl;        // row lanes per block = 256 / cg
    long rows_per_block;
    long xs, gs, ds;  // backward passes: elements between consecutive rows of x, dy and dx (C for dense [M, C] rows; wider when the
                      // rows are a channel slice of a wider channels-last tensor)
};

// ---- forward statistics ------------------------------------------------------------------------------------------------
// All threads of a block shift by the block's FIRST row (K[c] = x[r0][c]), so their shifted sums add directly; the
// block then emits (mean_b, M2_b) per channel.  partial layout per block: mean[C], m2[C] (floats); the row count of a
// block is implied by its index.
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const T* __restrict__ x, long m, int c, Geom g,
                                                            float* __restrict__ partial) {
    constexpr int V = Vec<T>::V;
    __shared__ float s_1[kThreads][V + 1];
    __shared__ float s_2[kThreads][V + 1];
    x += (size_t)blockIdx.y * m * c;                        // blockIdx.y = group (InstanceNorm: sample), 0 for BatchNorm
    partial += (size_t)blockIdx.y * gridDim.x * 2 * c;
    const int tid = threadIdx.x;
    const int col = tid % g.cg, rlane = tid / g.cg;
    const long r0 = (long)blockIdx.x * g.rows_per_block;
    const long r1 = r0 + g.rows_per_block < m ? r0 + g.rows_per_block : m;
    float K[V], s1[V], s2[V];
    Vec<T>::load(x + r0 * c + col * V, K);
#pragma unroll
    for (int j = 0; j < V; j++) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll 8
    for (long r = rlane < g.rl ? r0 + rlane : r1; r < r1; r += g.rl) {  // (rlane >= rl: C / V does not divide the block, idle lanes)
        float v[V];
        Vec<T>::load(x + r * c + col * V, v);
#pragma unroll
        for (int j = 0; j < V; j++) { const float d = v[j] - K[j]; s1[j] += d; s2[j] = fmaf(d, d, s2[j]); }
    }
#pragma unroll
    for (int j = 0; j < V; j++) { s_1[tid][j] = s1[j]; s_2[tid][j] = s2[j]; }
    __syncthreads();
    // one thread per channel finishes the block: tid -> (col2, j2)
    if (tid < c) {
        const int col2 = tid / V, j2 = tid % V;
        float a = 0.f, b2 = 0.f;
        for (int q = 0; q < g.rl; q++) { a += s_1[q * g.cg + col2][j2]; b2 += s_2[q * g.cg + col2][j2]; }
        const float n = (float)(r1 - r0);
        // K of this channel: re-read the first row (L1/L2 hit)
        float Kv[V];
        Vec<T>::load(x + r0 * c + col2 * V, Kv);
        float kk = 0.f;
#pragma unroll
        for (int j = 0; j < V; j++) if (j == j2) kk = Kv[j];
        float* p = partial + (size_t)blockIdx.x * 2 * c;
        p[tid] = kk + a / n;
        p[c + tid] = fmaxf(b2 - a * a / n, 0.f);
    }
}

// merge block partials: mean = sum n_b mean_b / N ; M2 = sum (M2_b + n_b (mean_b - mean)^2)   (two passes over the tiny
// partial array, fp64, fixed order) -> scale | shift | mean | invstd, running stats
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, long m, int c,
                                                           long rows_per_block, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float momentum, float eps,
                                                           float* __restrict__ stats) {
    __shared__ double sh[1024];
    __shared__ double sh_mean[512];
    partial += (size_t)blockIdx.x * nblk * 2 * c;           // blockIdx.x = group
    stats += (size_t)blockIdx.x * 4 * c;
    const int tid = threadIdx.x;
    const int chunks = 1024 / c > 0 ? 1024 / c : 1;
    const int ch = tid % c, chunk = tid / c;
    const int per = (nblk + chunks - 1) / chunks;
    const int lo = chunk * per, hi = (chunk < chunks) ? (lo + per < nblk ? lo + per : nblk) : lo;
    const double last_n = (double)(m - (long)(nblk - 1) * rows_per_block);
    // the partial rows are read 8 at a time (independent loads in flight) and added in the original order
    double acc = 0.0;
    {
        int b = lo;
        for (; b + 8 <= hi; b += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = partial[(size_t)(b + j) * 2 * c + ch];
#pragma unroll
            for (int j = 0; j < 8; j++) acc += (b + j == nblk - 1 ? last_n : (double)rows_per_block) * (double)v[j];
        }
        for (; b < hi; b++) acc += (b == nblk - 1 ? last_n : (double)rows_per_block) * (double)partial[(size_t)b * 2 * c + ch];
    }
    sh[tid] = acc;
    __syncthreads();
    if (tid < c) {
        double t = 0.0;
        for (int q = 0; q < chunks; q++) t += sh[q * c + tid];
        sh_mean[tid] = t / (double)m;
    }
    __syncthreads();
    const double mean = sh_mean[ch];
    acc = 0.0;
    {
        int b = lo;
        for (; b + 8 <= hi; b += 8) {
            float v[8], q2[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { v[j] = partial[(size_t)(b + j) * 2 * c + ch]; q2[j] = partial[(size_t)(b + j) * 2 * c + c + ch]; }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const double d = (double)v[j] - mean;
                acc += (double)q2[j] + (b + j == nblk - 1 ? last_n : (double)rows_per_block) * d * d;
            }
        }
        for (; b < hi; b++) {
            const double d = (double)partial[(size_t)b * 2 * c + ch] - mean;
            acc += (double)partial[(size_t)b * 2 * c + c + ch] + (b == nblk - 1 ? last_n : (double)rows_per_block) * d * d;
        }
    }
    __syncthreads();
    sh[tid] = acc;
    __syncthreads();
    if (tid < c) {
        double m2 = 0.0;
        for (int q = 0; q < chunks; q++) m2 += sh[q * c + tid];
        const double cnt = (double)m;
        const double var = m2 / cnt;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        stats[tid] = (float)((double)gamma[tid] * invstd);
        stats[c + tid] = (float)((double)beta[tid] - mean * (double)gamma[tid] * invstd);
        stats[2 * c + tid] = (float)mean;
        stats[3 * c + tid] = (float)invstd;
        if (running_mean && cnt > 1.0) {
            running_mean[tid] = (1.f - momentum) * running_mean[tid] + momentum * (float)mean;
            running_var[tid] = (1.f - momentum) * running_var[tid] + momentum * (float)(m2 / (cnt - 1.0));
        }
    }
}

__global__ void bn_eval_stats_kernel(int c, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float eps, float* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float invstd = 1.f / sqrtf(running_var[i] + eps);
    stats[i] = gamma[i] * invstd;
    stats[c + i] = beta[i] - running_mean[i] * gamma[i] * invstd;
    stats[2 * c + i] = running_mean[i];
    stats[3 * c + i] = invstd;
}

template <typename T, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const T* __restrict__ x, long m, int c, Geom g,
                                                            const float* __restrict__ stats, T* __restrict__ y) {
    constexpr int V = Vec<T>::V;
    const int col = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
    x += (size_t)blockIdx.y * m * c; y += (size_t)blockIdx.y * m * c; stats += (size_t)blockIdx.y * 4 * c;
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; j++) { sc[j] = stats[col * V + j]; sh[j] = stats[c + col * V + j]; }
    const long stride = (long)gridDim.x * g.rl;
    for (long r = rlane < g.rl ? (long)blockIdx.x * g.rl + rlane : m; r < m; r += stride) {
        float v[V];
        Vec<T>::load(x + r * c + col * V, v);
#pragma unroll
        for (int j = 0; j < V; j++) {
            v[j] = fmaf(v[j], sc[j], sh[j]);
            if (RELU) v[j] = fmaxf(v[j], 0.f);
        }
        Vec<T>::store(y + r * c + col * V, v);
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------------
// `ticket` != nullptr: the LAST block to finish (device-scope counter, left at zero again) also runs the finalize step below on
// the partial sums of all blocks -- one launch less per layer (the separate finalize launch costs 5-6 us of stream time for < 1 us of
// work); release / acquire fences at device scope make the other blocks' partial sums visible across the XCDs' L2 caches.
// Partial sums that another block of the SAME launch reads: device-scope (write-through / L2-bypassing) relaxed atomics instead of
// plain accesses + __threadfence() -- a device-scope release fence on this 8-XCD part writes back the whole L2 of the XCD (measured:
// ~15 us per layer, 3x the launch it saves).
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct BwdFinal {
    unsigned* ticket;
    const float* gamma;
    int training;
    float *grad_gamma, *grad_beta, *coef;
};

__device__ __forceinline__ void bn_bwd_finalize_block(const float* partial, int nblk, long m, int c, const float* gamma,
                                                      const float* stats, int training, float* grad_gamma, float* grad_beta,
                                                      float* coef, double* sh_a, double* sh_b, int nthreads) {
    const int tid = threadIdx.x;
    const int chunks = nthreads / c > 0 ? nthreads / c : 1;
    const int ch = tid % c, chunk = tid / c;
    double a = 0.0, b = 0.0;
    if (chunk < chunks) {
        const int per = (nblk + chunks - 1) / chunks;
        const int lo = chunk * per, hi = lo + per < nblk ? lo + per : nblk;
        int q = lo;
        for (; q + 8 <= hi; q += 8) {  // 16 independent loads in flight, original summation order
            float va[8], vb[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { va[j] = ld_agent(partial + (size_t)(q + j) * 2 * c + ch); vb[j] = ld_agent(partial + (size_t)(q + j) * 2 * c + c + ch); }
#pragma unroll
            for (int j = 0; j < 8; j++) { a += (double)va[j]; b += (double)vb[j]; }
        }
        for (; q < hi; q++) { a += (double)ld_agent(partial + (size_t)q * 2 * c + ch); b += (double)ld_agent(partial + (size_t)q * 2 * c + c + ch); }
    }
    sh_a[tid] = a; sh_b[tid] = b;
    __syncthreads();
    if (tid < c) {
        a = 0.0; b = 0.0;
        for (int q = 0; q < chunks; q++) { a += sh_a[q * c + tid]; b += sh_b[q * c + tid]; }
        grad_beta[tid] = (float)a;
        grad_gamma[tid] = (float)b;
        coef[tid] = gamma[tid] * stats[3 * c + tid];
        coef[c + tid] = training ? (float)(a / (double)m) : 0.f;
        coef[2 * c + tid] = training ? (float)(b / (double)m) : 0.f;
    }
}

template <typename T, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x, long m,
                                                                 int c, Geom g, const float* __restrict__ stats,
                                                                 float* partial, BwdFinal fin) {
    constexpr int V = Vec<T>::V;
    __shared__ float s_a[kThreads][V + 1];
    __shared__ float s_b[kThreads][V + 1];
    x += (size_t)blockIdx.y * m * g.xs; dy += (size_t)blockIdx.y * m * g.gs; stats += (size_t)blockIdx.y * 4 * c;
    partial += (size_t)blockIdx.y * gridDim.x * 2 * c;
    const int tid = threadIdx.x;
    const int col = tid % g.cg, rlane = tid / g.cg;
    float sc[V], sh[V], mu[V], is[V], a[V], b[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        sc[j] = stats[col * V + j]; sh[j] = stats[c + col * V + j];
        mu[j] = stats[2 * c + col * V + j]; is[j] = stats[3 * c + col * V + j];
        a[j] = 0.f; b[j] = 0.f;
    }
    const long r0 = (long)blockIdx.x * g.rows_per_block;
    const long r1 = r0 + g.rows_per_block < m ? r0 + g.rows_per_block : m;
#pragma unroll 4
    for (long r = rlane < g.rl ? r0 + rlane : r1; r < r1; r += g.rl) {
        float vx[V], vg[V];
        Vec<T>::load(x + r * g.xs + col * V, vx);
        Vec<T>::load(dy + r * g.gs + col * V, vg);
#pragma unroll
        for (int j = 0; j < V; j++) {
            float dz = vg[j];
            if (RELU && !(fmaf(vx[j], sc[j], sh[j]) > 0.f)) dz = 0.f;  // ReLU mask recomputed from x
            a[j] += dz;
            b[j] = fmaf(dz, (vx[j] - mu[j]) * is[j], b[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < V; j++) { s_a[tid][j] = a[j]; s_b[tid][j] = b[j]; }
    __syncthreads();
    if (tid < c) {
        const int col2 = tid / V, j2 = tid % V;
        float sa = 0.f, sb = 0.f;
        for (int q = 0; q < g.rl; q++) { sa += s_a[q * g.cg + col2][j2]; sb += s_b[q * g.cg + col2][j2]; }
        float* p = partial + (size_t)blockIdx.x * 2 * c;
        if (fin.ticket == nullptr) {
            p[tid] = sa;
            p[c + tid] = sb;
        } else {
            st_agent(p + tid, sa);
            st_agent(p + c + tid, sb);
        }
    }
    if (fin.ticket == nullptr) return;
    __shared__ int s_last;
    // Hand-off to the last block (MI355X_MICROARCH.md "Correctness boundaries", inter-workgroup visibility): every storing wave drains
    // its stores (vmcnt(0)), the block's barrier, ONE lane's agent-scope release in front of the ticket, and in the last block an
    // agent-scope acquire behind it (one lane, then the barrier) before the plain loads of the other blocks' partial sums.  Round 4 had
    // a workgroup-scope fence and a relaxed ticket here: neither orders the other waves' stores for another CU / XCD.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    if (tid == 0) {
        unsigned* tk = fin.ticket + blockIdx.y;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_last = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (zero again for the next call)
        }
    }
    __syncthreads();
    if (!s_last) return;
    double* sh_a = reinterpret_cast<double*>(&s_a[0][0]);  // (kThreads x (V + 1) floats >= kThreads doubles for V >= 1 ... V = 4 | 8)
    double* sh_b = reinterpret_cast<double*>(&s_b[0][0]);
    __syncthreads();
    bn_bwd_finalize_block(partial, gridDim.x, m, c, fin.gamma, stats, fin.training,
                          fin.grad_gamma + (size_t)blockIdx.y * c, fin.grad_beta + (size_t)blockIdx.y * c,
                          fin.coef + (size_t)blockIdx.y * 3 * c, sh_a, sh_b, kThreads);
}

// sums -> grad_beta, grad_gamma and the three dx coefficients per channel: dx = A * (dz - B - xhat * Cc)
// blockIdx.x = group (InstanceNorm sample), blockIdx.y = channel segment of `cw` channels (cw = c: one block per group).  With 32-channel
// segments every thread merges nblk / 32 partial sums -- one round of loads instead of four to eight dependent ones at 128 / 256 channels
// (the launch sits between the reduction and the dx pass of EVERY layer: 6.1 us each before, 23 per detector step).
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, long m, int c, int cw,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ stats, int training,
                                                               float* __restrict__ grad_gamma, float* __restrict__ grad_beta,
                                                               float* __restrict__ coef) {
    __shared__ double sh_a[1024], sh_b[1024];
    partial += (size_t)blockIdx.x * nblk * 2 * c;
    stats += (size_t)blockIdx.x * 4 * c; coef += (size_t)blockIdx.x * 3 * c;
    grad_gamma += (size_t)blockIdx.x * c; grad_beta += (size_t)blockIdx.x * c;
    const int tid = threadIdx.x;
    const int chunks = 1024 / cw > 0 ? 1024 / cw : 1;
    const int lc = tid % cw, chunk = tid / cw;
    const int ch = blockIdx.y * cw + lc;
    double a = 0.0, b = 0.0;
    if (chunk < chunks) {
        const int per = (nblk + chunks - 1) / chunks;
        const int lo = chunk * per, hi = lo + per < nblk ? lo + per : nblk;
        int q = lo;
        for (; q + 8 <= hi; q += 8) {  // 16 independent loads in flight, summed in block order
            float va[8], vb[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { va[j] = partial[(size_t)(q + j) * 2 * c + ch]; vb[j] = partial[(size_t)(q + j) * 2 * c + c + ch]; }
#pragma unroll
            for (int j = 0; j < 8; j++) { a += (double)va[j]; b += (double)vb[j]; }
        }
        for (; q < hi; q++) { a += (double)partial[(size_t)q * 2 * c + ch]; b += (double)partial[(size_t)q * 2 * c + c + ch]; }
    }
    sh_a[tid] = a; sh_b[tid] = b;
    __syncthreads();
    // the chunk sums per channel in chunk order; 32 chunks (32-channel segments) as a fixed two-level tree: 4 runs of 8, then the 4 run sums
    if (chunks == 32) {
        double ra = 0.0, rb = 0.0;
        if (tid < 4 * cw) {
            const int run = tid / cw;
#pragma unroll
            for (int q = 0; q < 8; q++) { ra += sh_a[(run * 8 + q) * cw + lc]; rb += sh_b[(run * 8 + q) * cw + lc]; }
        }
        __syncthreads();
        if (tid < 4 * cw) { sh_a[tid] = ra; sh_b[tid] = rb; }
        __syncthreads();
    }
    if (tid < cw) {
        a = 0.0; b = 0.0;
        const int left = chunks == 32 ? 4 : chunks;
        for (int q = 0; q < left; q++) { a += sh_a[q * cw + tid]; b += sh_b[q * cw + tid]; }
        grad_beta[ch] = (float)a;
        grad_gamma[ch] = (float)b;
        coef[ch] = gamma[ch] * stats[3 * c + ch];
        coef[c + ch] = training ? (float)(a / (double)m) : 0.f;
        coef[2 * c + ch] = training ? (float)(b / (double)m) : 0.f;
    }
}

// channel segment of the finalize launch: 32 where the channel count allows and there are enough partial sums to spread
inline int finalize_segment(int c, int nblk) { return (c % 32 == 0 && c > 32 && nblk >= 64) ? 32 : c; }

// InstanceNorm: the affine parameters are shared by all samples -> one block walks the groups (samples) in order, writes every
// group's dx coefficients and the SUM of the per-sample parameter gradients (fixed order): no [groups, C] intermediate and no
// reduction launches behind the call
__global__ __launch_bounds__(1024) void in_bwd_finalize_sum_kernel(const float* __restrict__ partial, int groups, int nblk, long m, int c,
                                                                   const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                   float* __restrict__ grad_gamma, float* __restrict__ grad_beta,
                                                                   float* __restrict__ coef) {
    __shared__ double sh_a[1024], sh_b[1024];
    const int tid = threadIdx.x;
    const int chunks = 1024 / c > 0 ? 1024 / c : 1;
    const int ch = tid % c, chunk = tid / c;
    double sum_a = 0.0, sum_b = 0.0;
    for (int g = 0; g < groups; g++) {
        const float* part = partial + (size_t)g * nblk * 2 * c;
        double a = 0.0, b = 0.0;
        if (chunk < chunks) {
            const int per = (nblk + chunks - 1) / chunks;
            const int lo = chunk * per, hi = lo + per < nblk ? lo + per : nblk;
            int q = lo;
            for (; q + 8 <= hi; q += 8) {
                float va[8], vb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { va[j] = part[(size_t)(q + j) * 2 * c + ch]; vb[j] = part[(size_t)(q + j) * 2 * c + c + ch]; }
#pragma unroll
                for (int j = 0; j < 8; j++) { a += (double)va[j]; b += (double)vb[j]; }
            }
            for (; q < hi; q++) { a += (double)part[(size_t)q * 2 * c + ch]; b += (double)part[(size_t)q * 2 * c + c + ch]; }
        }
        __syncthreads();  // (the previous group's reads of sh_a / sh_b)
        sh_a[tid] = a; sh_b[tid] = b;
        __syncthreads();
        if (tid < c) {
            a = 0.0; b = 0.0;
            for (int q = 0; q < chunks; q++) { a += sh_a[q * c + tid]; b += sh_b[q * c + tid]; }
            // (the per-sample values rounded to fp32 first, then added: what `grad[groups, C].sum(0)` of the three-step form computes)
            sum_a += (double)(float)a; sum_b += (double)(float)b;
            float* cf = coef + (size_t)g * 3 * c;
            cf[tid] = gamma[tid] * stats[(size_t)g * 4 * c + 3 * c + tid];
            cf[c + tid] = (float)(a / (double)m);
            cf[2 * c + tid] = (float)(b / (double)m);
        }
    }
    if (tid < c) { grad_beta[tid] = (float)sum_a; grad_gamma[tid] = (float)sum_b; }
}

template <typename T, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x, long m, int c,
                                                             Geom g, const float* __restrict__ stats,
                                                             const float* __restrict__ coef, T* __restrict__ dx) {
    constexpr int V = Vec<T>::V;
    const int col = threadIdx.x % g.cg, rlane = threadIdx.x / g.cg;
    x += (size_t)blockIdx.y * m * g.xs; dy += (size_t)blockIdx.y * m * g.gs; dx += (size_t)blockIdx.y * m * g.ds;
    stats += (size_t)blockIdx.y * 4 * c; coef += (size_t)blockIdx.y * 3 * c;
    float sc[V], sh[V], mu[V], is[V], A[V], Bc[V], Cc[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int ch = col * V + j;
        sc[j] = stats[ch]; sh[j] = stats[c + ch]; mu[j] = stats[2 * c + ch]; is[j] = stats[3 * c + ch];
        A[j] = coef[ch]; Bc[j] = coef[c + ch]; Cc[j] = coef[2 * c + ch];
    }
    const long stride = (long)gridDim.x * g.rl;
    for (long r = rlane < g.rl ? (long)blockIdx.x * g.rl + rlane : m; r < m; r += stride) {
        float vx[V], vg[V];
        Vec<T>::load(x + r * g.xs + col * V, vx);
        Vec<T>::load(dy + r * g.gs + col * V, vg);
#pragma unroll
        for (int j = 0; j < V; j++) {
            float dz = vg[j];
            if (RELU && !(fmaf(vx[j], sc[j], sh[j]) > 0.f)) dz = 0.f;
            vg[j] = A[j] * (dz - Bc[j] - (vx[j] - mu[j]) * is[j] * Cc[j]);
        }
        Vec<T>::store(dx + r * g.ds + col * V, vg);
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

inline bool geom(int c, int v, long m, Geom* g, int* nblk) {
    if (c <= 0 || c % v != 0 || c > kThreads) return false;
    const int cg = c / v;
    if (cg > kThreads) return false;
    g->cg = cg;
    g->rl = kThreads / cg;
    // ~256 blocks (1 per CU) keeps the single-block finalize short; never fewer than kRowsPerBlock rows per block
    long rpb = (m + 255) / 256;
    if (rpb < kRowsPerBlock) rpb = kRowsPerBlock;
    long nb = (m + rpb - 1) / rpb;
    if (nb > kMaxBlocks) { rpb = (m + kMaxBlocks - 1) / kMaxBlocks; nb = (m + rpb - 1) / rpb; }
    g->rows_per_block = rpb;
    g->xs = g->gs = g->ds = c;
    *nblk = (int)(nb > 0 ? nb : 1);
    return true;
}

inline int stream_grid(long m, const Geom& g) {
    long nb = (m + g.rl - 1) / g.rl;
    return (int)(nb < 8192 ? (nb > 0 ? nb : 1) : 8192);
}

}  // namespace

extern "C" {

size_t liso_bn_workspace_bytes(int c) {
    if (c <= 0) return 0;
    return ((size_t)kMaxBlocks * 2 * c + 3 * (size_t)c) * sizeof(float);
}

int liso_bn_relu_fwd(const void* x, int is_bf16, long m, int c, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int training, int relu,
                     void* y, float* stats, void* workspace, size_t workspace_bytes, void* stream) {
    Geom g;
    int nblk;
    if (m < 0 || !geom(c, is_bf16 ? 8 : 4, m, &g, &nblk)) return LISO_EINVAL;
    if (!gamma || !beta || !running_mean || !running_var || !stats || !workspace || (m > 0 && (!x || !y))) return LISO_EINVAL;
    if (workspace_bytes < liso_bn_workspace_bytes(c)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    if (training) {
        if (m == 0) return LISO_EINVAL;
        if (is_bf16)
            bn_stats_kernel<__hip_bfloat16><<<nblk, kThreads, 0, st>>>((const __hip_bfloat16*)x, m, c, g, partial);
        else
            bn_stats_kernel<float><<<nblk, kThreads, 0, st>>>((const float*)x, m, c, g, partial);
        bn_finalize_kernel<<<1, 1024, 0, st>>>(partial, nblk, m, c, g.rows_per_block, gamma, beta, running_mean, running_var,
                                               momentum, eps, stats);
    } else {
        bn_eval_stats_kernel<<<(c + 255) / 256, 256, 0, st>>>(c, gamma, beta, running_mean, running_var, eps, stats);
    }
    if (m > 0) {
        const int grid = stream_grid(m, g);
#define LISO_APPLY(T, R) bn_apply_kernel<T, R><<<grid, kThreads, 0, st>>>((const T*)x, m, c, g, stats, (T*)y)
        if (is_bf16) { if (relu) LISO_APPLY(__hip_bfloat16, true); else LISO_APPLY(__hip_bfloat16, false); }
        else { if (relu) LISO_APPLY(float, true); else LISO_APPLY(float, false); }
#undef LISO_APPLY
    }
    return check_launch();
}

static int bn_relu_bwd(const void* dy, const void* x, int is_bf16, long m, int c, const float* gamma, const float* stats,
                       int training, int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace,
                       size_t workspace_bytes, unsigned* ticket, void* stream, long dy_stride = 0, long x_stride = 0, long dx_stride = 0) {
    Geom g;
    int nblk;
    if (m <= 0 || !geom(c, is_bf16 ? 8 : 4, m, &g, &nblk)) return LISO_EINVAL;
    if (dy_stride || x_stride || dx_stride) {  // rows that are channel slices of wider channels-last tensors
        const int v = is_bf16 ? 8 : 4;
        if (dy_stride < c || x_stride < c || dx_stride < c || dy_stride % v || x_stride % v || dx_stride % v) return LISO_EINVAL;
        if ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) != 0) return LISO_EINVAL;
        g.gs = dy_stride; g.xs = x_stride; g.ds = dx_stride;
    }
    if (!dy || !x || !gamma || !stats || !dx || !grad_gamma || !grad_beta || !workspace) return LISO_EINVAL;
    if (workspace_bytes < liso_bn_workspace_bytes(c)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* coef = partial + (size_t)kMaxBlocks * 2 * c;
    const int grid = stream_grid(m, g);
    const BwdFinal fin{ticket, gamma, training, grad_gamma, grad_beta, coef};
    const int cw = finalize_segment(c, nblk);
#define LISO_BWD(T, R)                                                                                                     \
    do {                                                                                                                   \
        bn_bwd_reduce_kernel<T, R><<<nblk, kThreads, 0, st>>>((const T*)dy, (const T*)x, m, c, g, stats, partial, fin);     \
        if (!ticket)                                                                                                       \
            bn_bwd_finalize_kernel<<<dim3(1, c / cw), 1024, 0, st>>>(partial, nblk, m, c, cw, gamma, stats, training, grad_gamma,  \
                                                                     grad_beta, coef);                                    \
        bn_bwd_dx_kernel<T, R><<<grid, kThreads, 0, st>>>((const T*)dy, (const T*)x, m, c, g, stats, coef, (T*)dx);          \
    } while (0)
    if (is_bf16) { if (relu) LISO_BWD(__hip_bfloat16, true); else LISO_BWD(__hip_bfloat16, false); }
    else { if (relu) LISO_BWD(float, true); else LISO_BWD(float, false); }
#undef LISO_BWD
    return check_launch();
}

int liso_bn_relu_bwd(const void* dy, const void* x, int is_bf16, long m, int c, const float* gamma, const float* stats,
                     int training, int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace,
                     size_t workspace_bytes, void* stream) {
    return bn_relu_bwd(dy, x, is_bf16, m, c, gamma, stats, training, relu, dx, grad_gamma, grad_beta, workspace, workspace_bytes,
                       nullptr, stream);
}

int liso_bn_relu_bwd_strided(const void* dy, long dy_stride, const void* x, long x_stride, int is_bf16, long m, int c, const float* gamma,
                             const float* stats, int training, int relu, void* dx, long dx_stride, float* grad_gamma, float* grad_beta,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (dy_stride <= 0 || x_stride <= 0 || dx_stride <= 0) return LISO_EINVAL;
    return bn_relu_bwd(dy, x, is_bf16, m, c, gamma, stats, training, relu, dx, grad_gamma, grad_beta, workspace, workspace_bytes,
                       nullptr, stream, dy_stride, x_stride, dx_stride);
}

int liso_bn_relu_bwd_ticket(const void* dy, const void* x, int is_bf16, long m, int c, const float* gamma, const float* stats,
                            int training, int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace,
                            size_t workspace_bytes, unsigned* ticket, void* stream) {
    if (!ticket) return LISO_EINVAL;
    return bn_relu_bwd(dy, x, is_bf16, m, c, gamma, stats, training, relu, dx, grad_gamma, grad_beta, workspace, workspace_bytes,
                       ticket, stream);
}

size_t liso_in_workspace_bytes(int groups, int c) {
    if (groups <= 0 || c <= 0) return 0;
    return (size_t)groups * liso_bn_workspace_bytes(c);
}

int liso_in_relu_fwd(const void* x, int is_bf16, int groups, long m, int c, const float* gamma, const float* beta, float eps, int relu,
                     void* y, float* stats, void* workspace, size_t workspace_bytes, void* stream) {
    Geom g;
    int nblk;
    if (groups < 1 || m <= 0 || !geom(c, is_bf16 ? 8 : 4, m, &g, &nblk)) return LISO_EINVAL;
    if (!gamma || !beta || !stats || !workspace || !x || !y) return LISO_EINVAL;
    if (workspace_bytes < liso_in_workspace_bytes(groups, c)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    const dim3 gs((unsigned)nblk, (unsigned)groups), ga((unsigned)stream_grid(m, g), (unsigned)groups);
    if (is_bf16)
        bn_stats_kernel<__hip_bfloat16><<<gs, kThreads, 0, st>>>((const __hip_bfloat16*)x, m, c, g, partial);
    else
        bn_stats_kernel<float><<<gs, kThreads, 0, st>>>((const float*)x, m, c, g, partial);
    bn_finalize_kernel<<<groups, 1024, 0, st>>>(partial, nblk, m, c, g.rows_per_block, gamma, beta, nullptr, nullptr, 0.f, eps, stats);
#define LISO_APPLY(T, R) bn_apply_kernel<T, R><<<ga, kThreads, 0, st>>>((const T*)x, m, c, g, stats, (T*)y)
    if (is_bf16) { if (relu) LISO_APPLY(__hip_bfloat16, true); else LISO_APPLY(__hip_bfloat16, false); }
    else { if (relu) LISO_APPLY(float, true); else LISO_APPLY(float, false); }
#undef LISO_APPLY
    return check_launch();
}

static int in_relu_bwd(const void* dy, const void* x, int is_bf16, int groups, long m, int c, const float* gamma, const float* stats,
                     int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes, int summed, void* stream) {
    Geom g;
    int nblk;
    if (groups < 1 || m <= 0 || !geom(c, is_bf16 ? 8 : 4, m, &g, &nblk)) return LISO_EINVAL;
    if (!dy || !x || !gamma || !stats || !dx || !grad_gamma || !grad_beta || !workspace) return LISO_EINVAL;
    if (workspace_bytes < liso_in_workspace_bytes(groups, c)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* coef = partial + (size_t)groups * kMaxBlocks * 2 * c;
    const dim3 gs((unsigned)nblk, (unsigned)groups), ga((unsigned)stream_grid(m, g), (unsigned)groups);
    const int cw = finalize_segment(c, nblk);
#define LISO_BWD(T, R)                                                                                                     \
    do {                                                                                                                   \
        bn_bwd_reduce_kernel<T, R><<<gs, kThreads, 0, st>>>((const T*)dy, (const T*)x, m, c, g, stats, partial, BwdFinal{}); \
        if (summed)                                                                                                        \
            in_bwd_finalize_sum_kernel<<<1, 1024, 0, st>>>(partial, groups, nblk, m, c, gamma, stats, grad_gamma, grad_beta, coef); \
        else                                                                                                               \
            bn_bwd_finalize_kernel<<<dim3(groups, c / cw), 1024, 0, st>>>(partial, nblk, m, c, cw, gamma, stats, 1, grad_gamma,     \
                                                                          grad_beta, coef);                               \
        bn_bwd_dx_kernel<T, R><<<ga, kThreads, 0, st>>>((const T*)dy, (const T*)x, m, c, g, stats, coef, (T*)dx);            \
    } while (0)
    if (is_bf16) { if (relu) LISO_BWD(__hip_bfloat16, true); else LISO_BWD(__hip_bfloat16, false); }
    else { if (relu) LISO_BWD(float, true); else LISO_BWD(float, false); }
#undef LISO_BWD
    return check_launch();
}

int liso_in_relu_bwd(const void* dy, const void* x, int is_bf16, int groups, long m, int c, const float* gamma, const float* stats,
                     int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes, void* stream) {
    return in_relu_bwd(dy, x, is_bf16, groups, m, c, gamma, stats, relu, dx, grad_gamma, grad_beta, workspace, workspace_bytes, 0, stream);
}

int liso_in_relu_bwd_sum(const void* dy, const void* x, int is_bf16, int groups, long m, int c, const float* gamma, const float* stats,
                         int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes, void* stream) {
    return in_relu_bwd(dy, x, is_bf16, groups, m, c, gamma, stats, relu, dx, grad_gamma, grad_beta, workspace, workspace_bytes, 1, stream);
}

}  // extern "C"
