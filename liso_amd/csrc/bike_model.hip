// Kinematic bicycle-model rollout of a batch of tracks and its adjoint for gfx950.  C ABI + reference lines: include/liso_tracking.h.
//
// One lane per track walks the time axis (the recurrence is sequential in time and independent across tracks); the reference's
// scripted loop issues ~30 elementwise launches per time step and direction.  States live in HBM ([B,T,5], the forward result the
// loss reads anyway); the adjoint recomputes the two soft clamps' arguments from the stored states.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_tracking.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;

// soft_sigmoid_clamp (track_smoothing.py:23-35): a_min + (a_max - a_min) * (0.5 + atan(x / 100) / pi)
__device__ __forceinline__ float soft_clamp(float x, float lo, float hi) { return lo + (hi - lo) * (0.5f + 1.f / kPi * atanf(x / 100.f)); }
__device__ __forceinline__ float soft_clamp_grad(float x, float lo, float hi) {
    const float u = x / 100.f;
    return (hi - lo) * (1.f / kPi) / (1.f + u * u) / 100.f;
}

__global__ void bike_rollout_fwd_kernel(int batch, int steps, const float* __restrict__ init, const float* __restrict__ accel,
                                        const float* __restrict__ steer, const float* __restrict__ length, float dt, float max_yaw_rate,
                                        float max_velocity, float* __restrict__ states) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    float x = init[5 * b], y = init[5 * b + 1], h = init[5 * b + 2], v = init[5 * b + 3], hd = init[5 * b + 4];
    const float len = length[b];
    float* s = states + (size_t)b * steps * 5;
    s[0] = x; s[1] = y; s[2] = h; s[3] = v; s[4] = hd;
    for (int t = 0; t + 1 < steps; t++) {  // car_dynamics (:300-337)
        const float nhd = soft_clamp(hd + steer[(size_t)b * steps + t] * dt, -max_yaw_rate, max_yaw_rate);
        const float nh = h + dt * fabsf(v) / len * nhd;
        const float nv = soft_clamp(v + accel[(size_t)b * steps + t] * dt, 0.f, max_velocity);
        y = y + nv * sinf(nh) * dt;
        x = x + nv * cosf(nh) * dt;
        h = nh; v = nv; hd = nhd;
        float* o = s + 5 * (t + 1);
        o[0] = x; o[1] = y; o[2] = h; o[3] = v; o[4] = hd;
    }
}

__global__ void bike_rollout_bwd_kernel(int batch, int steps, const float* __restrict__ accel, const float* __restrict__ steer,
                                        const float* __restrict__ length, float dt, float max_yaw_rate, float max_velocity,
                                        const float* __restrict__ states, const float* __restrict__ gstates, float* __restrict__ ginit,
                                        float* __restrict__ gaccel, float* __restrict__ gsteer) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const float len = length[b];
    const float* s = states + (size_t)b * steps * 5;
    const float* g = gstates + (size_t)b * steps * 5;
    // gradient of the last time step's inputs: accel / steer of the last frame never enter the dynamics
    gaccel[(size_t)b * steps + steps - 1] = 0.f;
    gsteer[(size_t)b * steps + steps - 1] = 0.f;
    float gx = g[5 * (steps - 1)], gy = g[5 * (steps - 1) + 1], gh = g[5 * (steps - 1) + 2], gv = g[5 * (steps - 1) + 3],
          ghd = g[5 * (steps - 1) + 4];
    for (int t = steps - 2; t >= 0; t--) {
        const float h = s[5 * t + 2], v = s[5 * t + 3], hd = s[5 * t + 4];
        const float nh = s[5 * (t + 1) + 2], nv = s[5 * (t + 1) + 3], nhd = s[5 * (t + 1) + 4];
        const float sn = sinf(nh), cs = cosf(nh);
        // x' = x + nv cos(nh) dt ; y' = y + nv sin(nh) dt
        float g_nv = gv + gx * cs * dt + gy * sn * dt;
        float g_nh = gh + (-gx * nv * sn + gy * nv * cs) * dt;
        float g_nhd = ghd + g_nh * dt * fabsf(v) / len;
        // nv = clamp(v + a dt)
        const float dv = soft_clamp_grad(v + accel[(size_t)b * steps + t] * dt, 0.f, max_velocity);
        gaccel[(size_t)b * steps + t] = g_nv * dv * dt;
        // nhd = clamp(hd + steer dt)
        const float dh = soft_clamp_grad(hd + steer[(size_t)b * steps + t] * dt, -max_yaw_rate, max_yaw_rate);
        gsteer[(size_t)b * steps + t] = g_nhd * dh * dt;
        const float sign_v = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
        const float pgx = gx, pgy = gy;
        gx = g[5 * t] + pgx;
        gy = g[5 * t + 1] + pgy;
        gh = g[5 * t + 2] + g_nh;
        gv = g[5 * t + 3] + g_nv * dv + g_nh * dt * sign_v / len * nhd;
        ghd = g[5 * t + 4] + g_nhd * dh;
        (void)h;
    }
    ginit[5 * b] = gx; ginit[5 * b + 1] = gy; ginit[5 * b + 2] = gh; ginit[5 * b + 3] = gv; ginit[5 * b + 4] = ghd;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

}  // namespace

extern "C" {

int liso_bike_rollout_fwd_f32(int batch, int timesteps, const float* initial_state, const float* accel, const float* steering,
                              const float* vehicle_length, float dt, float max_yaw_rate, float max_velocity, float* states, void* stream) {
    if (batch < 0 || timesteps < 1) return LISO_EINVAL;
    if (batch == 0) return LISO_OK;
    if (!initial_state || !accel || !steering || !vehicle_length || !states) return LISO_EINVAL;
    bike_rollout_fwd_kernel<<<(batch + 63) / 64, 64, 0, (hipStream_t)stream>>>(batch, timesteps, initial_state, accel, steering,
                                                                             vehicle_length, dt, max_yaw_rate, max_velocity, states);
    return check_launch();
}

int liso_bike_rollout_bwd_f32(int batch, int timesteps, const float* accel, const float* steering, const float* vehicle_length, float dt,
                              float max_yaw_rate, float max_velocity, const float* states, const float* grad_states,
                              float* grad_initial_state, float* grad_accel, float* grad_steering, void* stream) {
    if (batch < 0 || timesteps < 1) return LISO_EINVAL;
    if (batch == 0) return LISO_OK;
    if (!accel || !steering || !vehicle_length || !states || !grad_states || !grad_initial_state || !grad_accel || !grad_steering)
        return LISO_EINVAL;
    bike_rollout_bwd_kernel<<<(batch + 63) / 64, 64, 0, (hipStream_t)stream>>>(batch, timesteps, accel, steering, vehicle_length, dt,
                                                                             max_yaw_rate, max_velocity, states, grad_states,
                                                                             grad_initial_state, grad_accel, grad_steering);
    return check_launch();
}

}  // extern "C"
