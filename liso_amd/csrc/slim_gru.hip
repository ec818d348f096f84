// ConvGRU gate arithmetic of the RAFT update block for gfx950 (elementwise, HBM/launch-bound).  C ABI + reference lines:
// include/liso_slim.h.  All maps are NCHW fp32; `*_bs` are batch strides in elements (the z / r pre-activations may be the two
// channel halves of one merged convolution output).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// z = sigmoid(cz); rhx[:, :Ch] = sigmoid(cr) * h; rhx[:, Ch:] = x          (update.py:31-34)
__global__ void gru_in_fwd_kernel(liso_gru_cfg c, const float* __restrict__ cz, const float* __restrict__ cr, long zr_bs,
                                  const float* __restrict__ h, const float* __restrict__ x, float* __restrict__ z,
                                  float* __restrict__ rhx) {
    const long per_b = (long)(c.ch + c.cx) * c.hw;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)c.batch * per_b) return;
    const long b = i / per_b, r = i - b * per_b;
    if (r < (long)c.ch * c.hw) {
        const float hv = h[b * c.ch * c.hw + r];
        z[b * c.ch * c.hw + r] = sigmoidf(cz[b * zr_bs + r]);
        rhx[i] = sigmoidf(cr[b * zr_bs + r]) * hv;
    } else {
        rhx[i] = x[b * c.cx * c.hw + (r - (long)c.ch * c.hw)];
    }
}

// adjoint: g_rhx -> g_cr = g_rh * h * r (1 - r), g_h = g_rh * r, (g_x is the tail of g_rhx: a view on the host side);
// g_z (gradient arriving at z from the output gate) -> g_cz = g_z * z (1 - z)
__global__ void gru_in_bwd_kernel(liso_gru_cfg c, const float* __restrict__ cr, long zr_bs, const float* __restrict__ h,
                                  const float* __restrict__ z, const float* __restrict__ g_z, const float* __restrict__ g_rhx,
                                  float* __restrict__ g_cz, float* __restrict__ g_cr, long gzr_bs, float* __restrict__ g_h) {
    const long per_b = (long)c.ch * c.hw;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)c.batch * per_b) return;
    const long b = i / per_b, r = i - b * per_b;
    const float rv = sigmoidf(cr[b * zr_bs + r]), hv = h[i], zv = z[i];
    const float grh = g_rhx[b * (long)(c.ch + c.cx) * c.hw + r];
    g_cr[b * gzr_bs + r] = grh * hv * rv * (1.f - rv);
    g_h[i] = grh * rv;
    g_cz[b * gzr_bs + r] = (g_z != nullptr ? g_z[i] : 0.f) * zv * (1.f - zv);
}

// h' = (1 - z) * h + z * tanh(cq)                                           (update.py:35-37)
__global__ void gru_out_fwd_kernel(long n, const float* __restrict__ cq, const float* __restrict__ z, const float* __restrict__ h,
                                   float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float zv = z[i];
    out[i] = (1.f - zv) * h[i] + zv * tanhf(cq[i]);
}

__global__ void gru_out_bwd_kernel(long n, const float* __restrict__ cq, const float* __restrict__ z, const float* __restrict__ h,
                                   const float* __restrict__ g_out, float* __restrict__ g_cq, float* __restrict__ g_z,
                                   float* __restrict__ g_h) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float zv = z[i], q = tanhf(cq[i]), g = g_out[i];
    g_cq[i] = g * zv * (1.f - q * q);
    g_z[i] = g * (q - h[i]);
    g_h[i] = g * (1.f - zv);
}

// ---- inference on pixel rows that are channel slices of wider channels-last buffers --------------------------------------------------
// (the update block's inputs [h | inp | out | class | flow | r*h] live in ONE buffer at inference: no concatenation passes)
__global__ void gru_in_rows_kernel(long n_pix, int ch, const float* __restrict__ zr, long zr_stride, const float* __restrict__ h,
                                   long h_stride, float* __restrict__ z, float* __restrict__ rh, long rh_stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * ch) return;
    const long p = i / ch;
    const int c = (int)(i - p * ch);
    z[i] = sigmoidf(zr[p * zr_stride + c]);
    rh[p * rh_stride + c] = sigmoidf(zr[p * zr_stride + ch + c]) * h[p * h_stride + c];
}

__global__ void gru_out_rows_kernel(long n_pix, int ch, const float* __restrict__ cq, long cq_stride, const float* __restrict__ z,
                                    float* __restrict__ h, long h_stride) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * ch) return;
    const long p = i / ch;
    const int c = (int)(i - p * ch);
    const float zv = z[i], hv = h[p * h_stride + c];
    h[p * h_stride + c] = (1.f - zv) * hv + zv * tanhf(cq[p * cq_stride + c]);
}

inline bool bad(const liso_gru_cfg* c) { return c == nullptr || c->batch <= 0 || c->ch <= 0 || c->cx < 0 || c->hw <= 0; }
inline int done() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

}  // namespace

extern "C" int liso_gru_in_fwd_f32(const liso_gru_cfg* c, const float* cz, const float* cr, long zr_batch_stride, const float* h,
                                   const float* x, float* z, float* rhx, void* stream) {
    if (bad(c) || !cz || !cr || !h || (c->cx > 0 && !x) || !z || !rhx) return LISO_EINVAL;
    const long n = (long)c->batch * (c->ch + c->cx) * c->hw;
    hipLaunchKernelGGL(gru_in_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *c, cz, cr,
                       zr_batch_stride, h, x, z, rhx);
    return done();
}

extern "C" int liso_gru_in_bwd_f32(const liso_gru_cfg* c, const float* cr, long zr_batch_stride, const float* h, const float* z,
                                   const float* g_z, const float* g_rhx, float* g_cz, float* g_cr, long gzr_batch_stride,
                                   float* g_h, void* stream) {
    if (bad(c) || !cr || !h || !z || !g_rhx || !g_cz || !g_cr || !g_h) return LISO_EINVAL;
    const long n = (long)c->batch * c->ch * c->hw;
    hipLaunchKernelGGL(gru_in_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *c, cr,
                       zr_batch_stride, h, z, g_z, g_rhx, g_cz, g_cr, gzr_batch_stride, g_h);
    return done();
}

extern "C" int liso_gru_out_fwd_f32(long n, const float* cq, const float* z, const float* h, float* out, void* stream) {
    if (n < 0 || (n > 0 && (!cq || !z || !h || !out))) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    hipLaunchKernelGGL(gru_out_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, cq, z, h, out);
    return done();
}

extern "C" int liso_gru_out_bwd_f32(long n, const float* cq, const float* z, const float* h, const float* g_out, float* g_cq,
                                    float* g_z, float* g_h, void* stream) {
    if (n < 0 || (n > 0 && (!cq || !z || !h || !g_out || !g_cq || !g_z || !g_h))) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    hipLaunchKernelGGL(gru_out_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, cq, z, h, g_out,
                       g_cq, g_z, g_h);
    return done();
}

extern "C" int liso_gru_in_rows_f32(long n_pix, int ch, const float* zr, long zr_stride, const float* h, long h_stride, float* z, float* rh,
                                    long rh_stride, void* stream) {
    if (n_pix < 0 || ch <= 0 || zr_stride < 2 * ch || h_stride < ch || rh_stride < ch || (n_pix > 0 && (!zr || !h || !z || !rh)))
        return LISO_EINVAL;
    if (n_pix == 0) return LISO_OK;
    const long n = n_pix * ch;
    hipLaunchKernelGGL(gru_in_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_pix, ch, zr, zr_stride,
                       h, h_stride, z, rh, rh_stride);
    return done();
}

extern "C" int liso_gru_out_rows_f32(long n_pix, int ch, const float* cq, long cq_stride, const float* z, float* h, long h_stride,
                                     void* stream) {
    if (n_pix < 0 || ch <= 0 || cq_stride < ch || h_stride < ch || (n_pix > 0 && (!cq || !z || !h))) return LISO_EINVAL;
    if (n_pix == 0) return LISO_OK;
    const long n = n_pix * ch;
    hipLaunchKernelGGL(gru_out_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n_pix, ch, cq, cq_stride,
                       z, h, h_stride);
    return done();
}

// ---- RAFT loop state at inference: one launch instead of `coords1 + d_flow`, `logits + d_logits`, `coords1 - coords0` ---------------------
// (liso/slim/model/raft.py:199-216: coords1 = coords1 + delta_flow; logits = logits + delta_logits; next iteration: flow = coords1 - coords0)
// The same three fp32 operations per pixel, in the same order; the results land where the next iteration reads them: coords1 [B,2,hw] for
// the correlation lookup, (flow | logits | 0 0) as ONE channels-last pixel of 8 floats for the motion encoder's merged 7x7 convolution.
namespace {
__global__ __launch_bounds__(256) void raft_state_step_kernel(int batch, int hw, const float* __restrict__ delta, long delta_stride,
                                                              const float* __restrict__ coords0, float* __restrict__ coords1,
                                                              float* __restrict__ state8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * hw) return;
    const int b = (int)(i / hw), p = (int)(i - (long)b * hw);
    const float* dl = delta + i * delta_stride;
    const long cx = ((long)b * 2) * hw + p, cy = cx + hw;
    const float c1x = coords1[cx] + dl[0], c1y = coords1[cy] + dl[1];
    coords1[cx] = c1x;
    coords1[cy] = c1y;
    float4* s = reinterpret_cast<float4*>(state8 + i * 8);
    const float4 lo = s[0], hi = s[1];
    s[0] = make_float4(c1x - coords0[cx], c1y - coords0[cy], lo.z + dl[2], lo.w + dl[3]);
    s[1] = make_float4(hi.x + dl[4], hi.y + dl[5], 0.0f, 0.0f);
}
}  // namespace

extern "C" int liso_raft_state_step_f32(int batch, int hw, const float* delta, long delta_stride, const float* coords0, float* coords1,
                                        float* state8, void* stream) {
    if (batch < 0 || hw < 0 || delta_stride < 6) return LISO_EINVAL;
    const long n = (long)batch * hw;
    if (n == 0) return LISO_OK;
    if (!delta || !coords0 || !coords1 || !state8 || ((uintptr_t)state8 & 15)) return LISO_EINVAL;
    hipLaunchKernelGGL(raft_state_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, batch, hw, delta,
                       delta_stride, coords0, coords1, state8);
    return done();
}
