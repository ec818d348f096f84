// Elementwise kernels of the RAFT update loop's TRAINING step on gfx950 (HBM / launch-bound).  C ABI + reference lines:
// include/liso_slim.h (liso_rows_combine_f32, liso_gru_*_rows_train_*, liso_raft_state_step_train_f32, liso_raft_pack_output_grads_f32).
//
// liso_amd/slim/model/raft_loop.py runs all RAFT iterations (liso/slim/model/raft.py:178-259 + update.py:29-164) as ONE autograd node:
// every activation of every iteration lives in a few stacked channels-last buffers ([iteration * batch, h, w, C]; the convolutions write
// their channel ranges, liso_conv.h), the backward pass is sequenced by hand, and the weight gradients are one launch per layer over the
// stacked iterations.  What is left between the convolutions are these kernels: they read and write PIXEL ROWS that are channel slices
// of wider buffers (strides in floats between consecutive pixels), 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }
inline int done() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }
inline bool row_ok(const void* p, long stride, int c) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && stride % 4 == 0 && stride >= c); }

// out[p, :] = (a[p, :] + b[p, :] + c[p, :]) * (mask[p, :] > 0)  (+ out[p, :] when `accumulate`); b, c, mask may be absent
__global__ __launch_bounds__(256) void rows_combine_kernel(long n_pix, int c4n, const float* __restrict__ a, long a_s, const float* __restrict__ b,
                                                           long b_s, const float* __restrict__ c, long c_s, const float* __restrict__ mask,
                                                           long m_s, float* __restrict__ out, long o_s, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * c4n) return;
    const long p = i / c4n;
    const int k = (int)(i - p * c4n) * 4;
    float4 v = *reinterpret_cast<const float4*>(a + p * a_s + k);
    if (b) {
        const float4 w = *reinterpret_cast<const float4*>(b + p * b_s + k);
        v = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
    }
    if (c) {
        const float4 w = *reinterpret_cast<const float4*>(c + p * c_s + k);
        v = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
    }
    if (mask) {
        const float4 m = *reinterpret_cast<const float4*>(mask + p * m_s + k);
        v = make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
    }
    float4* o = reinterpret_cast<float4*>(out + p * o_s + k);
    if (accumulate) {
        const float4 w = *o;
        v = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
    }
    *o = v;
}

// h_out = (1 - z) h_in + z tanh(cq)                                                             (update.py:35-37), out of place
__global__ __launch_bounds__(256) void gru_out_rows_train_kernel(long n_pix, int ch, const float* __restrict__ cq, long cq_s,
                                                                 const float* __restrict__ z, const float* __restrict__ h_in, long hi_s,
                                                                 float* __restrict__ h_out, long ho_s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * ch) return;
    const long p = i / ch;
    const int c = (int)(i - p * ch);
    const float zv = z[i];
    h_out[p * ho_s + c] = (1.f - zv) * h_in[p * hi_s + c] + zv * tanhf(cq[p * cq_s + c]);
}

// adjoint of it (the arithmetic of gru_out_bwd_kernel, slim_gru.hip): g_cq = g z (1 - q^2), g_z = g (q - h), g_h = g (1 - z)
__global__ __launch_bounds__(256) void gru_out_rows_bwd_kernel(long n_pix, int ch, const float* __restrict__ cq, long cq_s,
                                                               const float* __restrict__ z, const float* __restrict__ h_in, long hi_s,
                                                               const float* __restrict__ g, long g_s, float* __restrict__ g_cq, long gcq_s,
                                                               float* __restrict__ g_z, float* __restrict__ g_h) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * ch) return;
    const long p = i / ch;
    const int c = (int)(i - p * ch);
    const float zv = z[i], q = tanhf(cq[p * cq_s + c]), gv = g[p * g_s + c];
    g_cq[p * gcq_s + c] = gv * zv * (1.f - q * q);
    g_z[i] = gv * (q - h_in[p * hi_s + c]);
    g_h[i] = gv * (1.f - zv);
}

// adjoint of gru_in_rows (z = sigmoid(zr[:ch]), rh = sigmoid(zr[ch:]) h): g_zr[:ch] = g_z z (1 - z), g_zr[ch:] = g_rh h r (1 - r),
// g_h = g_rh r   (the arithmetic of gru_in_bwd_kernel)
__global__ __launch_bounds__(256) void gru_in_rows_bwd_kernel(long n_pix, int ch, const float* __restrict__ zr, long zr_s,
                                                              const float* __restrict__ h, long h_s, const float* __restrict__ z,
                                                              const float* __restrict__ g_z, const float* __restrict__ g_rh, long grh_s,
                                                              float* __restrict__ g_zr, long gzr_s, float* __restrict__ g_h) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * ch) return;
    const long p = i / ch;
    const int c = (int)(i - p * ch);
    const float rv = sigmoidf(zr[p * zr_s + ch + c]), hv = h[p * h_s + c], zv = z[i], grh = g_rh[p * grh_s + c];
    g_zr[p * gzr_s + ch + c] = grh * hv * rv * (1.f - rv);
    g_h[i] = grh * rv;
    g_zr[p * gzr_s + c] = g_z[i] * zv * (1.f - zv);
}

// The loop's state update (raft.py:199-216) with every iteration's state kept: state pixel = (logit 0..3, flow x, flow y, 0, 0).
//   coords_out = coords_in + delta[4:6];  state_out = (state_in[0:4] + delta[0:4], coords_out - coords0, 0, 0)
// and the same numbers once more in the planar layout the output assembly reads (flow [b, 2, hw], logits [b, 4, hw]).
__global__ __launch_bounds__(256) void raft_state_step_train_kernel(int batch, int hw, const float* __restrict__ delta8,
                                                                    const float* __restrict__ coords0, const float* __restrict__ coords_in,
                                                                    float* __restrict__ coords_out, const float* __restrict__ state_in,
                                                                    float* __restrict__ state_out, float* __restrict__ flow_out,
                                                                    float* __restrict__ logits_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * hw) return;
    const int b = (int)(i / hw), p = (int)(i - (long)b * hw);
    const float4 dl = *reinterpret_cast<const float4*>(delta8 + i * 8), df = *reinterpret_cast<const float4*>(delta8 + i * 8 + 4);
    const float4 sl = *reinterpret_cast<const float4*>(state_in + i * 8);
    const long cx = ((long)b * 2) * hw + p, cy = cx + hw;
    const float c1x = coords_in[cx] + df.x, c1y = coords_in[cy] + df.y;
    coords_out[cx] = c1x;
    coords_out[cy] = c1y;
    const float4 l = make_float4(sl.x + dl.x, sl.y + dl.y, sl.z + dl.z, sl.w + dl.w);
    const float fx = c1x - coords0[cx], fy = c1y - coords0[cy];
    *reinterpret_cast<float4*>(state_out + i * 8) = l;
    *reinterpret_cast<float4*>(state_out + i * 8 + 4) = make_float4(fx, fy, 0.f, 0.f);
    flow_out[cx] = fx;
    flow_out[cy] = fy;
    const long lb = ((long)b * 4) * hw + p;
    logits_out[lb] = l.x;
    logits_out[lb + hw] = l.y;
    logits_out[lb + 2L * hw] = l.z;
    logits_out[lb + 3L * hw] = l.w;
}

// planar output gradients (g_flow [n, 2, hw], g_logits [n, 4, hw]; n = iterations x batch) -> one pixel of 8 floats (logits | flow | 0 0):
// the gradient of the heads' merged output convolution, in its own layout
__global__ __launch_bounds__(256) void raft_pack_output_grads_kernel(long n, int hw, const float* __restrict__ g_flow,
                                                                     const float* __restrict__ g_logits, float* __restrict__ g8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * hw) return;
    const long b = i / hw;
    const int p = (int)(i - b * hw);
    const float* gl = g_logits + b * 4 * hw + p;
    const float* gf = g_flow + b * 2 * hw + p;
    *reinterpret_cast<float4*>(g8 + i * 8) = make_float4(gl[0], gl[hw], gl[2L * hw], gl[3L * hw]);
    *reinterpret_cast<float4*>(g8 + i * 8 + 4) = make_float4(gf[0], gf[hw], 0.f, 0.f);
}

}  // namespace

extern "C" {

int liso_rows_combine_f32(long n_pix, int channels, const float* a, long a_stride, const float* b, long b_stride, const float* c,
                          long c_stride, const float* mask, long mask_stride, float* out, long out_stride, int accumulate, void* stream) {
    if (n_pix < 0 || channels <= 0 || channels % 4) return LISO_EINVAL;
    if (n_pix == 0) return LISO_OK;
    if (!a || !out || !row_ok(a, a_stride, channels) || !row_ok(b, b_stride, channels) || !row_ok(c, c_stride, channels) ||
        !row_ok(mask, mask_stride, channels) || !row_ok(out, out_stride, channels))
        return LISO_EINVAL;
    const long n = n_pix * (channels / 4);
    rows_combine_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n_pix, channels / 4, a, a_stride, b, b_stride, c, c_stride,
                                                                                    mask, mask_stride, out, out_stride, accumulate);
    return done();
}

int liso_gru_out_rows_train_f32(long n_pix, int ch, const float* cq, long cq_stride, const float* z, const float* h_in, long h_in_stride,
                                float* h_out, long h_out_stride, void* stream) {
    if (n_pix < 0 || ch <= 0 || cq_stride < ch || h_in_stride < ch || h_out_stride < ch || (n_pix > 0 && (!cq || !z || !h_in || !h_out)))
        return LISO_EINVAL;
    if (n_pix == 0) return LISO_OK;
    const long n = n_pix * ch;
    gru_out_rows_train_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n_pix, ch, cq, cq_stride, z, h_in, h_in_stride, h_out,
                                                                                          h_out_stride);
    return done();
}

int liso_gru_out_rows_bwd_f32(long n_pix, int ch, const float* cq, long cq_stride, const float* z, const float* h_in, long h_in_stride,
                              const float* g_out, long g_out_stride, float* g_cq, long g_cq_stride, float* g_z, float* g_h, void* stream) {
    if (n_pix < 0 || ch <= 0 || cq_stride < ch || h_in_stride < ch || g_out_stride < ch || g_cq_stride < ch ||
        (n_pix > 0 && (!cq || !z || !h_in || !g_out || !g_cq || !g_z || !g_h)))
        return LISO_EINVAL;
    if (n_pix == 0) return LISO_OK;
    const long n = n_pix * ch;
    gru_out_rows_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n_pix, ch, cq, cq_stride, z, h_in, h_in_stride, g_out,
                                                                                        g_out_stride, g_cq, g_cq_stride, g_z, g_h);
    return done();
}

int liso_gru_in_rows_bwd_f32(long n_pix, int ch, const float* zr, long zr_stride, const float* h, long h_stride, const float* z,
                             const float* g_z, const float* g_rh, long g_rh_stride, float* g_zr, long g_zr_stride, float* g_h, void* stream) {
    if (n_pix < 0 || ch <= 0 || zr_stride < 2 * ch || h_stride < ch || g_rh_stride < ch || g_zr_stride < 2 * ch ||
        (n_pix > 0 && (!zr || !h || !z || !g_z || !g_rh || !g_zr || !g_h)))
        return LISO_EINVAL;
    if (n_pix == 0) return LISO_OK;
    const long n = n_pix * ch;
    gru_in_rows_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n_pix, ch, zr, zr_stride, h, h_stride, z, g_z, g_rh,
                                                                                       g_rh_stride, g_zr, g_zr_stride, g_h);
    return done();
}

int liso_raft_state_step_train_f32(int batch, int hw, const float* delta8, const float* coords0, const float* coords_in, float* coords_out,
                                   const float* state_in, float* state_out, float* flow_out, float* logits_out, void* stream) {
    if (batch < 0 || hw < 0) return LISO_EINVAL;
    const long n = (long)batch * hw;
    if (n == 0) return LISO_OK;
    if (!delta8 || !coords0 || !coords_in || !coords_out || !state_in || !state_out || !flow_out || !logits_out ||
        ((((uintptr_t)delta8) | ((uintptr_t)state_in) | ((uintptr_t)state_out)) & 15))
        return LISO_EINVAL;
    raft_state_step_train_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(batch, hw, delta8, coords0, coords_in, coords_out,
                                                                                             state_in, state_out, flow_out, logits_out);
    return done();
}

int liso_raft_pack_output_grads_f32(long n, int hw, const float* g_flow, const float* g_logits, float* g8, void* stream) {
    if (n < 0 || hw < 0) return LISO_EINVAL;
    if (n * hw == 0) return LISO_OK;
    if (!g_flow || !g_logits || !g8 || (((uintptr_t)g8) & 15)) return LISO_EINVAL;
    const long t = n * hw;
    raft_pack_output_grads_kernel<<<(unsigned)((t + 255) / 256), 256, 0, (hipStream_t)stream>>>(n, hw, g_flow, g_logits, g8);
    return done();
}

}  // extern "C"
