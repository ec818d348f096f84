// AdamW over one flat fp32 parameter buffer on gfx950.  C ABI + reference lines: include/liso_optim.h.
//
// HBM-bound: 16 B read (p, g, m, v) + 12 B written (p, m, v) per element, nothing else.  Every thread moves one float4 of
// each stream per iteration (a wave reads 1 KiB contiguous per stream); the grid is sized to a few waves per SIMD and
// strides over the buffer.  The per-element operation order follows torch's multi-tensor AdamW (see the header), so the
// trajectories agree with torch.optim.AdamW to rounding.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_optim.h"

namespace {

struct AdamwScalars {
    float decay;      // 1 - lr * weight_decay
    float w1;         // 1 - beta1
    float beta2, w2;  // beta2, 1 - beta2
    float bc2_sqrt;   // sqrt(1 - beta2^step)
    float eps;
    float step_size;  // lr / (1 - beta1^step)
    float gscale;     // factor on the gradient (1 / world size behind a SUM all-reduce; 1 = none, bit-identical to no factor)
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamwScalars& s) {
    g = g * s.gscale;
    p = p * s.decay;
    m = fmaf(s.w1, g - m, m);
    v = fmaf(s.w2, g * g, v * s.beta2);
    const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
    p = fmaf(-s.step_size, m / denom, p);
}

__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, size_t n4, size_t n, AdamwScalars s) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adamw_one(pp.x, gg.x, mm.x, vv.x, s);
        adamw_one(pp.y, gg.y, mm.y, vv.y, s);
        adamw_one(pp.z, gg.z, mm.z, vv.z, s);
        adamw_one(pp.w, gg.w, mm.w, vv.w, s);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0) {  // tail (n % 4 elements)
        const size_t i = 4 * n4 + threadIdx.x;
        if (i < n) adamw_one(p[i], g[i], m[i], v[i], s);
    }
}

// ---- RMSprop over one flat buffer (SLIM's optimizer: liso/slim/experiment.py:200-219, torch.optim.RMSprop defaults) ---------------------
// torch's multi-tensor form, per element: sq = sq * alpha + (1 - alpha) * g * g;  p = p - lr * g / (sqrt(sq) + eps)
// HBM-bound: 12 B read (p, g, sq) + 8 B written (p, sq) per element, one launch instead of five foreach launches per 1-3 chunks.
__global__ __launch_bounds__(256) void rmsprop_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq,
                                                           size_t n4, size_t n, float lr, float alpha, float w, float eps, float gscale) {
    auto one = [&](float& pp, float gg, float& ss) {
        gg = gg * gscale;
        ss = fmaf(w, gg * gg, ss * alpha);
        pp = fmaf(-lr, gg / (sqrtf(ss) + eps), pp);
    };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 ss = reinterpret_cast<float4*>(sq)[i];
        one(pp.x, gg.x, ss.x);
        one(pp.y, gg.y, ss.y);
        one(pp.z, gg.z, ss.z);
        one(pp.w, gg.w, ss.w);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(sq)[i] = ss;
    }
    if (blockIdx.x == 0) {
        const size_t i = 4 * n4 + threadIdx.x;
        if (i < n) one(p[i], g[i], sq[i]);
    }
}

// ---- gradients that autograd produced outside the flat buffer: one launch moves all of them into their slices ----------------------
struct GatherTable {
    const float* src[LISO_GATHER_MAX];
    float* dst[LISO_GATHER_MAX];
    unsigned n[LISO_GATHER_MAX];
};

__global__ __launch_bounds__(256) void gather_f32_kernel(GatherTable t) {
    const float* __restrict__ s = t.src[blockIdx.y];
    float* __restrict__ d = t.dst[blockIdx.y];
    const unsigned n = t.n[blockIdx.y];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

}  // namespace

extern "C" int liso_gather_f32(int count, const void* const* src, void* const* dst, const size_t* numel, void* stream) {
    if (count < 0 || (count > 0 && (!src || !dst || !numel))) return LISO_EINVAL;
    for (int base = 0; base < count; base += LISO_GATHER_MAX) {
        GatherTable t;
        const int m = count - base < LISO_GATHER_MAX ? count - base : LISO_GATHER_MAX;
        size_t longest = 0;
        for (int k = 0; k < m; k++) {
            if (!src[base + k] || !dst[base + k] || numel[base + k] > 0xffffffffull) return LISO_EINVAL;
            t.src[k] = (const float*)src[base + k];
            t.dst[k] = (float*)dst[base + k];
            t.n[k] = (unsigned)numel[base + k];
            longest = numel[base + k] > longest ? numel[base + k] : longest;
        }
        size_t bx = (longest + 1023) / 1024;  // (4 elements per thread and pass)
        bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
        hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)bx, (unsigned)m), dim3(256), 0, (hipStream_t)stream, t);
    }
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

extern "C" int liso_adamw_step_scaled_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                                          double beta1, double beta2, double eps, double weight_decay, double grad_scale, long step,
                                          void* stream);

extern "C" int liso_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                                   double beta1, double beta2, double eps, double weight_decay, long step, void* stream) {
    return liso_adamw_step_scaled_f32(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, 1.0, step, stream);
}

extern "C" int liso_adamw_step_scaled_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr,
                                          double beta1, double beta2, double eps, double weight_decay, double grad_scale, long step,
                                          void* stream) {
    if (n == 0) return LISO_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) return LISO_EINVAL;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) return LISO_EINVAL;
    AdamwScalars s;
    s.decay = (float)(1.0 - lr * weight_decay);
    s.w1 = (float)(1.0 - beta1);
    s.beta2 = (float)beta2;
    s.w2 = (float)(1.0 - beta2);
    s.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    s.eps = (float)eps;
    s.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    s.gscale = (float)grad_scale;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 blocks: the rest is the grid-stride loop
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n4, n, s);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

extern "C" int liso_rmsprop_step_f32(float* param, const float* grad, float* square_avg, size_t n, double lr, double alpha, double eps,
                                     double grad_scale, void* stream) {
    if (n == 0) return LISO_OK;
    if (!param || !grad || !square_avg) return LISO_EINVAL;
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)square_avg) & 15) != 0) return LISO_EINVAL;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rmsprop_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, square_avg, n4, n,
                       (float)lr, (float)alpha, (float)(1.0 - alpha), (float)eps, (float)grad_scale);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}
