// Minimum-jerk track smoothing for gfx950: all Adam iterations of smooth_track_jerk in ONE launch.
// C ABI + reference lines: include/liso_tracking.h (liso_smooth_tracks_jerk_f32).
//
// The reference (liso/tracker/track_smoothing.py:104-230) runs 2000 Adam steps on a [B, T, 3] position tensor through autograd: ~25
// tiny launches per step, 1.8 s per batch on the host, for a problem of a few thousand floats.  Here one block owns one track: the
// positions and the normalised third differences live in LDS, Adam's moments in registers, and an iteration is two barrier-separated
// phases (jerk directions; gradient + update).  The gradient is the analytic one of
//   L = mean_b [ sum_t valid[b,t] * |p[t+3] - 3 p[t+2] + 3 p[t+1] - p[t]| / n_b  +  w * sum_t valid[b,t] * |p[t] - obs[t]|^2 / n_b ]
// (the third difference is zero-padded at the END of the track, so frame t is masked by valid[t] while it reads p[t..t+3]; padded
// frames are free parameters exactly as in the reference; p[0] is fixed).  fp32 throughout, torch.optim.Adam's operation order.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_tracking.h"

namespace {

constexpr int kMaxT = 1024;

__global__ __launch_bounds__(kMaxT) void smooth_tracks_jerk_kernel(const float* __restrict__ obs, const uint8_t* __restrict__ valid, int B,
                                                                   int T, int iters, float lr, float w_reg, float* __restrict__ out) {
    extern __shared__ float lds[];  // p[T][3] | u[T][3]
    float* p = lds;
    float* u = lds + 3 * T;
    __shared__ int n_valid_s;
    const int b = blockIdx.x, t = threadIdx.x;
    const float* ob = obs + (long)b * T * 3;
    const uint8_t* vb = valid + (long)b * T;
    if (t == 0) n_valid_s = 0;
    __syncthreads();
    const bool act = t < T;
    float o[3] = {0.f, 0.f, 0.f};
    float mask = 0.f;
    if (act) {
        for (int c = 0; c < 3; c++) { o[c] = ob[t * 3 + c]; p[t * 3 + c] = o[c]; }
        mask = vb[t] ? 1.f : 0.f;
        if (vb[t]) atomicAdd(&n_valid_s, 1);
    }
    __syncthreads();
    const float inv_bn = 1.0f / ((float)B * (float)n_valid_s);  // d(mean over tracks of sum / n_b)
    float m1[3] = {0.f, 0.f, 0.f}, m2[3] = {0.f, 0.f, 0.f};
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    float b1p = 1.f, b2p = 1.f;  // beta^step
    for (int it = 0; it < iters; it++) {
        // ---- phase 1: u[t] = valid[t] * d / |d|,  d = third forward difference at t (t <= T - 4) -----------------------------------
        if (act) {
            float d[3] = {0.f, 0.f, 0.f};
            if (t + 3 < T) {
                for (int c = 0; c < 3; c++)
                    d[c] = ((p[(t + 3) * 3 + c] - 3.f * p[(t + 2) * 3 + c]) + 3.f * p[(t + 1) * 3 + c]) - p[t * 3 + c];
            }
            const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const float s = nrm > 0.f ? mask / nrm : 0.f;  // (torch's norm backward is 0 at 0)
            for (int c = 0; c < 3; c++) u[t * 3 + c] = d[c] * s;
        }
        __syncthreads();
        // ---- phase 2: gradient of frame t, Adam step (p[0] is not a parameter) --------------------------------------------------------
        b1p *= beta1;
        b2p *= beta2;
        if (act && t >= 1) {
            const float bc1 = 1.f - b1p, bc2_sqrt = sqrtf(1.f - b2p);
            const float step_size = lr / bc1;
            for (int c = 0; c < 3; c++) {
                float gj = -u[t * 3 + c];                       // coefficient -1 of frame t in its own difference
                if (t >= 1) gj += 3.f * u[(t - 1) * 3 + c];     // +3 in the difference that starts one frame earlier
                if (t >= 2) gj -= 3.f * u[(t - 2) * 3 + c];
                if (t >= 3) gj += u[(t - 3) * 3 + c];
                const float g = (gj + w_reg * 2.f * mask * (p[t * 3 + c] - o[c])) * inv_bn;
                m1[c] = beta1 * m1[c] + (1.f - beta1) * g;
                m2[c] = beta2 * m2[c] + (1.f - beta2) * g * g;
                const float denom = sqrtf(m2[c]) / bc2_sqrt + eps;
                p[t * 3 + c] = p[t * 3 + c] - step_size * (m1[c] / denom);
            }
        }
        __syncthreads();
    }
    if (act)
        for (int c = 0; c < 3; c++) out[((long)b * T + t) * 3 + c] = p[t * 3 + c];
}

}  // namespace

extern "C" int liso_smooth_tracks_jerk_f32(const float* observed_pos, const uint8_t* valid, int batch, int timesteps, int max_iters,
                                           float learning_rate, float pos_regul_loss_weight, float* smooth_pos, void* stream) {
    if (batch < 0 || timesteps < 1 || timesteps > kMaxT || max_iters < 0) return LISO_EINVAL;
    if (batch == 0) return LISO_OK;
    if (!observed_pos || !valid || !smooth_pos) return LISO_EINVAL;
    int threads = 64;
    while (threads < timesteps) threads <<= 1;
    const size_t lds = (size_t)6 * timesteps * sizeof(float);
    smooth_tracks_jerk_kernel<<<batch, threads, lds, (hipStream_t)stream>>>(observed_pos, valid, batch, timesteps, max_iters, learning_rate,
                                                                           pos_regul_loss_weight, smooth_pos);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}
