// Nearest-point flow loss for gfx950: warped point -> gathered nearest neighbour -> squared distance -> field-of-view
// mask -> (huber) norm, forward and backward, one thread per point.  C ABI + reference lines: include/liso_slim.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

struct Eval {
    float d2, loss, dx, dy, dz, scale;  // scale: d loss / d (q - nn) = scale * (q - nn)
};

__device__ __forceinline__ Eval evaluate(const liso_nploss_cfg& c, const float* __restrict__ cloud_a, const float* __restrict__ flow,
                                         const float* __restrict__ cloud_b, const long long* __restrict__ idx, long row) {
    Eval e;
    const long b = row / c.n;
    const float qx = cloud_a[3 * row] + flow[3 * row], qy = cloud_a[3 * row + 1] + flow[3 * row + 1],
                qz = cloud_a[3 * row + 2] + flow[3 * row + 2];
    const long j = (long)idx[row];
    const float* nn = cloud_b + ((size_t)b * c.n_b + (j >= 0 && j < c.n_b ? j : 0)) * 3;
    e.dx = nn[0] - qx; e.dy = nn[1] - qy; e.dz = nn[2] - qz;            // knn_wrapper.py:199-203: nearest - cloud_b__a
    e.d2 = e.dx * e.dx + e.dy * e.dy + e.dz * e.dz;
    // field of view (:82-115)
    const float min_fov = fminf(fminf(qx - c.ext[0], qy - c.ext[1]), fminf(c.ext[2] - qx, c.ext[3] - qy));
    float w = 1.f;
    if (c.fov_mode == 1) w = min_fov > 0.f ? 1.f : 0.f;                                    // ignore_out_fov
    else if (c.fov_mode == 2) w = (min_fov > 0.f && e.d2 < min_fov * min_fov) ? 1.f : 0.f;  // mask_close_fov
    // huber_delta(err_sqr, delta, "large_grad_1") (:11-49); delta == 0: gradient-safe sqrt
    float l, dl_dd2;
    if (c.delta == 0.f) {
        const bool nz = !(e.d2 == 0.f);
        l = nz ? sqrtf(e.d2) : 0.f;
        dl_dd2 = nz ? 0.5f / l : 0.f;
    } else {
        const float dd = c.delta * c.delta;
        l = fminf(e.d2, dd) / (2.f * c.delta) + sqrtf(fmaxf(e.d2, dd)) - c.delta;
        dl_dd2 = e.d2 < dd ? 1.f / (2.f * c.delta) : (e.d2 > dd ? 0.5f / sqrtf(e.d2) : 1.f / c.delta);  // both clamps pass at ==
    }
    e.loss = l * w;
    e.scale = -2.f * dl_dd2 * w;  // d d2 / d q = -2 (nn - q)
    if (!(e.d2 == e.d2)) {  // NaN row (padding): torch propagates NaN through clamp / sqrt / NaN * 0; fminf / fmaxf would not
        e.loss = nanf("");
        e.scale = 0.f;
    }
    return e;
}

__global__ void nploss_fwd_kernel(liso_nploss_cfg c, const float* __restrict__ cloud_a, const float* __restrict__ flow,
                                  const float* __restrict__ cloud_b, const long long* __restrict__ idx, float* __restrict__ loss,
                                  float* __restrict__ dist_sqr) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (long)c.batch * c.n) return;
    const Eval e = evaluate(c, cloud_a, flow, cloud_b, idx, row);
    loss[row] = e.loss;       // NaN rows (padding marked with NaN by the caller) stay NaN, as in the reference
    dist_sqr[row] = e.d2;
}

__global__ void nploss_bwd_kernel(liso_nploss_cfg c, const float* __restrict__ cloud_a, const float* __restrict__ flow,
                                  const float* __restrict__ cloud_b, const long long* __restrict__ idx,
                                  const float* __restrict__ g_loss, const float* __restrict__ g_dist_sqr, float* __restrict__ g_flow) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= (long)c.batch * c.n) return;
    const Eval e = evaluate(c, cloud_a, flow, cloud_b, idx, row);
    float s = g_loss[row] * e.scale;
    if (g_dist_sqr) s += g_dist_sqr[row] * -2.f;
    const bool ok = isfinite(e.d2);
    g_flow[3 * row + 0] = ok ? s * e.dx : 0.f;
    g_flow[3 * row + 1] = ok ? s * e.dy : 0.f;
    g_flow[3 * row + 2] = ok ? s * e.dz : 0.f;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }
inline bool cfg_ok(const liso_nploss_cfg* c) {
    return c && c->batch >= 1 && c->n >= 0 && c->n_b >= 1 && c->fov_mode >= 0 && c->fov_mode <= 2 && c->delta >= 0.f;
}

}  // namespace

extern "C" {

int liso_nearest_point_loss_fwd_f32(const liso_nploss_cfg* cfg, const float* cloud_a, const float* flow, const float* cloud_b,
                                    const int64_t* index, float* loss, float* dist_sqr, void* stream) {
    if (!cfg_ok(cfg)) return LISO_EINVAL;
    const long rows = (long)cfg->batch * cfg->n;
    if (rows == 0) return LISO_OK;
    if (!cloud_a || !flow || !cloud_b || !index || !loss || !dist_sqr) return LISO_EINVAL;
    nploss_fwd_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, (hipStream_t)stream>>>(*cfg, cloud_a, flow, cloud_b,
                                                                                      (const long long*)index, loss, dist_sqr);
    return check_launch();
}

int liso_nearest_point_loss_bwd_f32(const liso_nploss_cfg* cfg, const float* cloud_a, const float* flow, const float* cloud_b,
                                    const int64_t* index, const float* grad_loss, const float* grad_dist_sqr, float* grad_flow,
                                    void* stream) {
    if (!cfg_ok(cfg)) return LISO_EINVAL;
    const long rows = (long)cfg->batch * cfg->n;
    if (rows == 0) return LISO_OK;
    if (!cloud_a || !flow || !cloud_b || !index || !grad_loss || !grad_flow) return LISO_EINVAL;
    nploss_bwd_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, (hipStream_t)stream>>>(*cfg, cloud_a, flow, cloud_b,
                                                                                      (const long long*)index, grad_loss,
                                                                                      grad_dist_sqr, grad_flow);
    return check_launch();
}

}  // extern "C"
