// RAFT output assembly for gfx950: x8 bilinear upsampling (align_corners) of the low-resolution flow / class-logit maps of
// ALL update iterations, flow convention change, channel concat and channels-last layout in one pass, plus the adjoint.
// C ABI + reference lines: include/liso_slim.h.  HBM-bound: the forward writes S*H*W*32 B once, the backward reads it once.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

// at::native area_pixel_compute_source_index(align_corners=true) + compute_source_index_and_lambda (UpSample.cuh)
struct Tap {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Tap tap(float scale, int dst, int in_size) {
    const float src = scale * dst;
    Tap t;
    t.i0 = (int)src;
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.l1 = src - t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}
__device__ __forceinline__ float scale_of(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (out_size - 1) : 0.f;
}
// output sample s = dir * (n_it * B) + it * B + b  <-  input sample it * (dirs * B) + dir * B + b
__device__ __forceinline__ int input_sample(const liso_upsample_cfg& c, int s) {
    const int B = c.batch2 / c.dirs, dir = s / (c.n_it * B), r = s % (c.n_it * B), it = r / B, b = r % B;
    return it * c.batch2 + dir * B + b;
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(liso_upsample_cfg c, const float* __restrict__ flow,
                                                           const float* __restrict__ logits, float* __restrict__ out) {
    const int H = c.h * c.factor, W = c.w * c.factor;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)c.n_it * c.batch2 * H * W;
    if (pix >= total) return;
    const int x = pix % W, y = (pix / W) % H, s = pix / ((long)W * H);
    const int in = input_sample(c, s);
    const Tap ty = tap(scale_of(c.h, H), y, c.h), tx = tap(scale_of(c.w, W), x, c.w);
    const size_t plane = (size_t)c.h * c.w;
    auto sample = [&](const float* base) {
        const float* r0 = base + (size_t)ty.i0 * c.w;
        const float* r1 = base + (size_t)ty.i1 * c.w;
        return ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1]);
    };
    float4 lo, fl;
    const float* lg = logits + (size_t)in * 4 * plane;
    lo.x = sample(lg); lo.y = sample(lg + plane); lo.z = sample(lg + 2 * plane); lo.w = sample(lg + 3 * plane);
    const float* fw = flow + (size_t)in * 2 * plane;
    const float fx = sample(fw) * c.flow_scale, fy = sample(fw + plane) * c.flow_scale;  // RAFT (x = col, y = row) pixels
    fl.x = fy; fl.y = fx; fl.z = fy; fl.w = fx;                                           // (row, col) metres, static | dynamic
    float4* o = (float4*)(out + (size_t)pix * 8);
    o[0] = lo;
    o[1] = fl;
}

// adjoint, pass 1: along x.  One block per high-resolution row: the row (W x 8 floats) goes through LDS, every thread
// produces low-resolution columns.  mid[s, y, j, 0:4] = logits, mid[s, y, j, 4:6] = (flow_x, flow_y) adjoints.
__global__ __launch_bounds__(256) void upsample_bwd_x_kernel(liso_upsample_cfg c, const float* __restrict__ grad_out,
                                                             float* __restrict__ mid) {
    extern __shared__ float row[];  // [W][8]
    const int W = c.w * c.factor;
    const long ry = blockIdx.x;  // s * H + y
    const float4* g = (const float4*)(grad_out + (size_t)ry * W * 8);
    for (int i = threadIdx.x; i < W * 2; i += blockDim.x) ((float4*)row)[i] = g[i];
    __syncthreads();
    const float sc = scale_of(c.w, W);
    for (int o = threadIdx.x; o < c.w * 6; o += blockDim.x) {
        const int j = o / 6, ch = o % 6;
        // high-resolution columns whose taps touch j: src in (j - 1, j + 1)
        int x_lo = sc > 0.f ? (int)floorf((j - 1) / sc) : 0, x_hi = sc > 0.f ? (int)ceilf((j + 1) / sc) : W - 1;
        x_lo = max(x_lo, 0);
        x_hi = min(x_hi, W - 1);
        float acc = 0.f;
        for (int x = x_lo; x <= x_hi; ++x) {
            const Tap t = tap(sc, x, c.w);
            float wgt = 0.f;
            if (t.i0 == j) wgt += t.l0;
            if (t.i1 == j) wgt += t.l1;
            if (wgt == 0.f) continue;
            const float* p = row + x * 8;
            const float v = ch < 4 ? p[ch] : (ch == 4 ? p[5] + p[7] : p[4] + p[6]);  // flow_x <- col channels, flow_y <- row
            acc += wgt * v;
        }
        mid[((size_t)ry * c.w + j) * 6 + ch] = acc;
    }
}

// adjoint, pass 2: along y.  One block per (sample, low-resolution row); consecutive threads = consecutive (j, channel).
__global__ __launch_bounds__(256) void upsample_bwd_y_kernel(liso_upsample_cfg c, const float* __restrict__ mid,
                                                             float* __restrict__ grad_flow, float* __restrict__ grad_logits) {
    const int H = c.h * c.factor;
    const int s = blockIdx.x / c.h, i = blockIdx.x % c.h;
    const int in = input_sample(c, s);
    const float sc = scale_of(c.h, H);
    int y_lo = sc > 0.f ? (int)floorf((i - 1) / sc) : 0, y_hi = sc > 0.f ? (int)ceilf((i + 1) / sc) : H - 1;
    y_lo = max(y_lo, 0);
    y_hi = min(y_hi, H - 1);
    const size_t plane = (size_t)c.h * c.w;
    for (int o = threadIdx.x; o < c.w * 6; o += blockDim.x) {
        const int j = o / 6, ch = o % 6;
        float acc = 0.f;
        for (int y = y_lo; y <= y_hi; ++y) {
            const Tap t = tap(sc, y, c.h);
            float wgt = 0.f;
            if (t.i0 == i) wgt += t.l0;
            if (t.i1 == i) wgt += t.l1;
            if (wgt == 0.f) continue;
            acc += wgt * mid[(((size_t)s * H + y) * c.w + j) * 6 + ch];
        }
        if (ch < 4) grad_logits[((size_t)in * 4 + ch) * plane + (size_t)i * c.w + j] = acc;
        else grad_flow[((size_t)in * 2 + (ch - 4)) * plane + (size_t)i * c.w + j] = acc * c.flow_scale;
    }
}

bool bad(const liso_upsample_cfg* c) {
    return c == nullptr || c->n_it <= 0 || c->batch2 <= 0 || c->dirs <= 0 || c->batch2 % c->dirs != 0 || c->h <= 0 || c->w <= 0 ||
           c->factor <= 0;
}

}  // namespace

extern "C" size_t liso_raft_upsample_scratch_bytes(const liso_upsample_cfg* c) {
    if (bad(c)) return 0;
    return (size_t)c->n_it * c->batch2 * c->h * c->factor * c->w * 6 * sizeof(float);
}

extern "C" int liso_raft_upsample_outputs_fwd_f32(const liso_upsample_cfg* c, const float* flow_lr, const float* logits_lr,
                                                  float* out, void* stream) {
    if (bad(c) || flow_lr == nullptr || logits_lr == nullptr || out == nullptr) return LISO_EINVAL;
    const long total = (long)c->n_it * c->batch2 * c->h * c->factor * c->w * c->factor;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *c, flow_lr,
                       logits_lr, out);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

extern "C" int liso_raft_upsample_outputs_bwd_f32(const liso_upsample_cfg* c, const float* grad_out, void* scratch,
                                                  size_t scratch_bytes, float* grad_flow_lr, float* grad_logits_lr, void* stream) {
    if (bad(c) || grad_out == nullptr || grad_flow_lr == nullptr || grad_logits_lr == nullptr) return LISO_EINVAL;
    if (scratch == nullptr || scratch_bytes < liso_raft_upsample_scratch_bytes(c)) return LISO_EWORKSPACE;
    const int S = c->n_it * c->batch2, H = c->h * c->factor, W = c->w * c->factor;
    const size_t lds = (size_t)W * 8 * sizeof(float);
    if (lds > 64 * 1024) return LISO_EINVAL;  // W <= 2048
    hipLaunchKernelGGL(upsample_bwd_x_kernel, dim3((unsigned)((size_t)S * H)), dim3(256), lds, (hipStream_t)stream, *c, grad_out,
                       (float*)scratch);
    hipLaunchKernelGGL(upsample_bwd_y_kernel, dim3((unsigned)(S * c->h)), dim3(256), 0, (hipStream_t)stream, *c,
                       (const float*)scratch, grad_flow_lr, grad_logits_lr);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}
