// Weighted first/second moments of two point clouds (the reductions of a weighted Kabsch fit) for gfx950, with the
// matching backward.  C ABI + reference lines: include/liso_kabsch.h.
//
//   moments_partial   256 blocks x 256 threads, grid-stride over the points; every thread keeps the 16 sums in fp64
//                     registers, waves reduce by butterfly, one fp64 row of 16 per block -> `partials`
//   moments_final     one wave adds the block rows in a fixed order -> out[16]      (no atomics: bit reproducible)
//   moments_bwd       elementwise: one thread per point
// The points are read exactly once (N * 28 B); everything after the 16 numbers is 3x3 algebra on the host stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_kabsch.h"

namespace {

constexpr int kBlocks = 256, kThreads = 256;

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(kThreads) void moments_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                   const float* __restrict__ w, long n,
                                                                   double* __restrict__ partials) {
    // blockIdx.y = sample of the batch: x, y [B, n, 3], w [B, n], partials [B, kBlocks, 16]
    x += (size_t)blockIdx.y * n * 3; y += (size_t)blockIdx.y * n * 3; w += (size_t)blockIdx.y * n;
    partials += (size_t)blockIdx.y * kBlocks * 16;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0.0;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)kBlocks * kThreads) {
        const double wi = (double)w[i];
        const double xs[3] = {(double)x[3 * i], (double)x[3 * i + 1], (double)x[3 * i + 2]};
        const double ys[3] = {(double)y[3 * i], (double)y[3 * i + 1], (double)y[3 * i + 2]};
        acc[0] += wi;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            acc[1 + a] += wi * xs[a];
            acc[4 + a] += wi * ys[a];
#pragma unroll
            for (int b = 0; b < 3; b++) acc[7 + 3 * a + b] += wi * ys[a] * xs[b];  // S_yx[a][b] = sum w y_a x_b
        }
    }
    __shared__ double red[kThreads / 64][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        double v = acc[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        double v = 0.0;
        for (int wv = 0; wv < kThreads / 64; wv++) v += red[wv][threadIdx.x];
        partials[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
    }
}

__global__ void moments_final_kernel(const double* __restrict__ partials, double* __restrict__ out) {
    if (threadIdx.x >= 16) return;
    partials += (size_t)blockIdx.x * kBlocks * 16;
    double v = 0.0;
    for (int b = 0; b < kBlocks; b += 8) {  // 8 loads in flight, fixed summation order
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = partials[(size_t)(b + j) * 16 + threadIdx.x];
#pragma unroll
        for (int j = 0; j < 8; j++) v += t[j];
    }
    out[(size_t)blockIdx.x * 16 + threadIdx.x] = v;
}

__global__ void moments_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ w, long n,
                                   const double* __restrict__ g, float* __restrict__ gx, float* __restrict__ gy,
                                   float* __restrict__ gw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    {   // blockIdx.y = sample
        const size_t o = (size_t)blockIdx.y * n;
        x += o * 3; y += o * 3; w += o; g += (size_t)blockIdx.y * 16;
        if (gx) gx += o * 3;
        if (gy) gy += o * 3;
        if (gw) gw += o;
    }
    const double wi = (double)w[i];
    const double xs[3] = {(double)x[3 * i], (double)x[3 * i + 1], (double)x[3 * i + 2]};
    const double ys[3] = {(double)y[3 * i], (double)y[3 * i + 1], (double)y[3 * i + 2]};
    double dw = g[0];
    double dx[3] = {g[1], g[2], g[3]}, dy[3] = {g[4], g[5], g[6]};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        dw += g[1 + a] * xs[a] + g[4 + a] * ys[a];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const double G = g[7 + 3 * a + b];
            dw += G * ys[a] * xs[b];
            dy[a] += G * xs[b];
            dx[b] += G * ys[a];
        }
    }
    if (gw) gw[i] = (float)dw;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (gx) gx[3 * i + a] = (float)(wi * dx[a]);
        if (gy) gy[3 * i + a] = (float)(wi * dy[a]);
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

}  // namespace

extern "C" {

size_t liso_weighted_moments_workspace_bytes(int batch) { return (size_t)(batch > 0 ? batch : 0) * kBlocks * 16 * sizeof(double); }

int liso_weighted_moments_fwd_f32(const float* x, const float* y, const float* w, int batch, long n, double* out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    if (batch < 1 || n < 0 || !out || !workspace || (n > 0 && (!x || !y || !w))) return LISO_EINVAL;
    if (workspace_bytes < liso_weighted_moments_workspace_bytes(batch)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    moments_partial_kernel<<<dim3(kBlocks, batch), kThreads, 0, st>>>(x, y, w, n, (double*)workspace);
    moments_final_kernel<<<batch, 64, 0, st>>>((const double*)workspace, out);
    return check_launch();
}

int liso_weighted_moments_bwd_f32(const float* x, const float* y, const float* w, int batch, long n, const double* grad_out,
                                  float* grad_x, float* grad_y, float* grad_w, void* stream) {
    if (batch < 1 || n < 0 || !grad_out || (n > 0 && (!x || !y || !w))) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    moments_bwd_kernel<<<dim3((unsigned)((n + 255) / 256), batch), 256, 0, (hipStream_t)stream>>>(x, y, w, n, grad_out, grad_x, grad_y,
                                                                                                 grad_w);
    return check_launch();
}

}  // extern "C"
