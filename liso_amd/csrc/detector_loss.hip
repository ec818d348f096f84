// Fused CenterPoint decode + loss for gfx950.  C ABI + reference lines: include/liso_detector.h.
//
//   centerloss_partial   one thread per output cell: activations (tanh, softplus), decode (cell centre + offset, z prior),
//                        the four loss integrands and the unit-circle regulariser -> 12 sums, reduced per block
//                        (wave butterflies, fixed order) into fp64 partial rows
//   centerloss_final     one wave adds the block rows in order and forms the weighted losses
//   centerloss_bwd       one thread per output cell: closed-form derivatives scaled by the normalisers of the forward
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_detector.h"
#include "../../include/liso_iou3d.h"

namespace {

constexpr int kThreads = 256;
constexpr int NS = LISO_CENTERLOSS_NSUM;
// sums: 0 focal_pos 1 focal_neg 2 num_pos 3 rot_num 4 rot_den 5 dims_sum 6 pos_sum 7 n_sel 8 reg_sum

struct Maps {
    const float *pos, *dims, *rot, *probs;
    long s[16];
};

struct Pixel {
    float tp[3], sp[3], r[2], logit;  // tanh(pos), softplus(dims), rot, logit
    float raw_dims[3];
    float dec_pos[3];
};

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch: beta 1, threshold 20
__device__ __forceinline__ float log_sigmoid_f(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sgn_f(float x) { return (x > 0.f) - (x < 0.f); }

__device__ __forceinline__ Pixel load_pixel(const liso_centerloss_cfg& c, const Maps& m, const float* __restrict__ centers, int b,
                                            int y, int x) {
    Pixel p;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        p.tp[k] = tanhf(m.pos[b * m.s[0] + k * m.s[1] + y * m.s[2] + x * m.s[3]]);
        p.raw_dims[k] = m.dims[b * m.s[4] + k * m.s[5] + y * m.s[6] + x * m.s[7]];
        p.sp[k] = softplus_f(p.raw_dims[k]);
    }
#pragma unroll
    for (int k = 0; k < 2; k++) p.r[k] = m.rot[b * m.s[8] + k * m.s[9] + y * m.s[10] + x * m.s[11]];
    p.logit = m.probs[b * m.s[12] + y * m.s[14] + x * m.s[15]];
    const float cx = centers[((size_t)y * c.w + x) * 2 + 0], cy = centers[((size_t)y * c.w + x) * 2 + 1];
    p.dec_pos[0] = cx + c.res_x * 0.5f * p.tp[0];
    p.dec_pos[1] = cy + c.res_y * 0.5f * p.tp[1];
    p.dec_pos[2] = c.z_min + 0.5f * (p.tp[2] + 1.0f) * (c.z_max - c.z_min);
    return p;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(kThreads) void centerloss_partial_kernel(liso_centerloss_cfg c, Maps m, const float* __restrict__ gt_probs,
                                                                      const float* __restrict__ gt_dims, const float* __restrict__ gt_pos,
                                                                      const float* __restrict__ gt_rot,
                                                                      const uint8_t* __restrict__ center_mask,
                                                                      const uint8_t* __restrict__ ignore_mask,
                                                                      const float* __restrict__ rot_weights,
                                                                      const float* __restrict__ centers, double* __restrict__ partials) {
    const long n = (long)c.batch * c.h * c.w;
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; k++) acc[k] = 0.0;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        const int b = (int)(i / ((long)c.h * c.w));
        const int rem = (int)(i - (long)b * c.h * c.w);
        const int y = rem / c.w, x = rem - y * c.w;
        const Pixel p = load_pixel(c, m, centers, b, y, x);
        const bool center = center_mask[i] != 0, ign = ignore_mask ? ignore_mask[i] != 0 : false;
        // CenterNet focal loss, alpha 0.5, gamma 2, beta 4 (centerpoint_loss.py:165-200)
        const float pp = sigmoid_f(p.logit), pn = sigmoid_f(-p.logit);
        if (!ign) {
            if (center) acc[0] += (double)(0.5f * pn * pn * log_sigmoid_f(p.logit));
            else {
                const float om = 1.0f - gt_probs[i];
                acc[1] += (double)(0.5f * pp * pp * (om * om) * (om * om) * log_sigmoid_f(-p.logit));
            }
        }
        if (center) acc[2] += 1.0;
        if (center && !ign) {
            const float w = fmaxf(rot_weights ? rot_weights[i] : 1.0f, 0.1f);  // :37-60
            acc[3] += (double)(w * (fabsf(p.r[0] - gt_rot[2 * i]) + fabsf(p.r[1] - gt_rot[2 * i + 1])));
            acc[4] += (double)w;
            acc[5] += (double)(fabsf(p.sp[0] - gt_dims[3 * i]) + fabsf(p.sp[1] - gt_dims[3 * i + 1]) + fabsf(p.sp[2] - gt_dims[3 * i + 2]));
            acc[6] += (double)(fabsf(p.dec_pos[0] - gt_pos[3 * i]) + fabsf(p.dec_pos[1] - gt_pos[3 * i + 1]) +
                               fabsf(p.dec_pos[2] - gt_pos[3 * i + 2]));
            acc[7] += 1.0;
        }
        const float len = sqrtf(p.r[0] * p.r[0] + p.r[1] * p.r[1]);
        acc[8] += (double)((len - 1.0f) * (len - 1.0f));
    }
    __shared__ double red[kThreads / 64][9];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += shfl_xor_f64(v, o);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double v = 0.0;
        for (int wv = 0; wv < kThreads / 64; wv++) v += red[wv][threadIdx.x];
        partials[(size_t)blockIdx.x * NS + threadIdx.x] = v;
    }
}

__global__ void centerloss_final_kernel(liso_centerloss_cfg c, const double* __restrict__ partials, int nblocks,
                                        double* __restrict__ sums, float* __restrict__ losses) {
    // one wave per sum: lane l adds the blocks l, l + 64, ... (independent loads), then a fixed shuffle tree -- the single thread per
    // sum of the first version walked 256 dependent fp64 loads: 50 us at B = 4, more than the loss kernel itself
    __shared__ double s[NS];
    const int which = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (which < NS) {
        double v = 0.0;
        if (which < 9)
            for (int b = lane; b < nblocks; b += 64) v += partials[(size_t)b * NS + which];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0) {
            s[which] = v;
            sums[which] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double num_pos = fmax(s[2], 1.0);
        const double l_probs = -(s[0] + s[1]) / num_pos;
        const double l_rot = 10.0 * s[3] / fmax(s[4], 1.0);
        const double l_dims = s[5] / fmax(s[7] * 3.0, 1.0) / num_pos;
        const double l_pos = s[6] / fmax(s[7] * 3.0, 1.0) / num_pos;
        const double l_reg = s[8] / ((double)c.batch * c.h * c.w);
        losses[0] = (float)l_probs; losses[1] = (float)l_rot; losses[2] = (float)l_dims; losses[3] = (float)l_pos;
        losses[4] = (float)l_reg;
        losses[5] = (float)((double)c.sup_weight * (l_probs + l_rot + l_dims + l_pos) + (double)c.rot_reg_weight * l_reg);
    }
}

__global__ __launch_bounds__(kThreads) void centerloss_bwd_kernel(liso_centerloss_cfg c, Maps m, const float* __restrict__ gt_probs,
                                                                  const float* __restrict__ gt_dims, const float* __restrict__ gt_pos,
                                                                  const float* __restrict__ gt_rot, const uint8_t* __restrict__ center_mask,
                                                                  const uint8_t* __restrict__ ignore_mask,
                                                                  const float* __restrict__ rot_weights, const float* __restrict__ centers,
                                                                  const double* __restrict__ sums, const float* __restrict__ grad_total,
                                                                  float* __restrict__ g_pos, float* __restrict__ g_dims,
                                                                  float* __restrict__ g_rot, float* __restrict__ g_probs) {
    const long n = (long)c.batch * c.h * c.w;
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int b = (int)(i / ((long)c.h * c.w));
    const int rem = (int)(i - (long)b * c.h * c.w);
    const int y = rem / c.w, x = rem - y * c.w;
    const Pixel p = load_pixel(c, m, centers, b, y, x);
    const bool center = center_mask[i] != 0, ign = ignore_mask ? ignore_mask[i] != 0 : false;
    const float g = grad_total[0] * c.sup_weight;
    const float num_pos = fmaxf((float)sums[2], 1.0f);
    const float n_el = fmaxf((float)sums[7] * 3.0f, 1.0f);
    // d probs
    float gl = 0.f;
    if (!ign) {
        const float pp = sigmoid_f(p.logit), pn = sigmoid_f(-p.logit);
        if (center) gl = 0.5f * (pn * pn * pn - 2.0f * pn * pn * pp * log_sigmoid_f(p.logit));
        else {
            const float om = 1.0f - gt_probs[i];
            gl = 0.5f * (om * om) * (om * om) * (2.0f * pp * pp * pn * log_sigmoid_f(-p.logit) - pp * pp * pp);
        }
        gl = -gl / num_pos;
    }
    g_probs[b * m.s[12] + y * m.s[14] + x * m.s[15]] = g * gl;
    const bool sel = center && !ign;
    // d rot: weighted L1 + unit-circle regulariser
    const float len = sqrtf(p.r[0] * p.r[0] + p.r[1] * p.r[1]);
    const float reg_scale = grad_total[0] * c.rot_reg_weight * 2.0f * (len - 1.0f) / ((float)n * fmaxf(len, 1e-30f));
    const float w = sel ? 10.0f * fmaxf(rot_weights ? rot_weights[i] : 1.0f, 0.1f) / fmaxf((float)sums[4], 1.0f) : 0.f;
#pragma unroll
    for (int k = 0; k < 2; k++)
        g_rot[b * m.s[8] + k * m.s[9] + y * m.s[10] + x * m.s[11]] =
            g * w * sgn_f(p.r[k] - gt_rot[2 * i + k]) + (len > 0.f ? reg_scale * p.r[k] : 0.f);
    const float cd = sel ? g / (n_el * num_pos) : 0.f;
    const float scale_pos[3] = {0.5f * c.res_x, 0.5f * c.res_y, 0.5f * (c.z_max - c.z_min)};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float ds = p.raw_dims[k] > 20.f ? 1.0f : sigmoid_f(p.raw_dims[k]);  // softplus'
        g_dims[b * m.s[4] + k * m.s[5] + y * m.s[6] + x * m.s[7]] = sel ? cd * sgn_f(p.sp[k] - gt_dims[3 * i + k]) * ds : 0.f;
        g_pos[b * m.s[0] + k * m.s[1] + y * m.s[2] + x * m.s[3]] =
            sel ? cd * sgn_f(p.dec_pos[k] - gt_pos[3 * i + k]) * scale_pos[k] * (1.0f - p.tp[k] * p.tp[k]) : 0.f;
    }
}

// ---- target rendering ----------------------------------------------------------------------------------------------------
// cell centres: get_voxel_center_coords_m (liso/utils/bev_utils.py:24-40): ((i + 0.5) / H) * range - range / 2, in fp64 then fp32
__device__ __forceinline__ float cell_center(int i, int n, float range) {
    return (float)((((double)i + 0.5) / (double)n) * (double)range - 0.5 * (double)range);
}

__device__ __forceinline__ float box_heat(float cx, float cy, float bx, float by, float c, float s, float len, float wid) {
    const float dx = cx - bx, dy = cy - by;
    const float u = dx * c + dy * s, v = -dx * s + dy * c;
    const float fac = u * u / (0.15f * len) + v * v / (0.15f * wid);  // kabsch_mask.py:93-102
    return expf(-fac / 2.0f);
}

__global__ __launch_bounds__(256) void targets_box_max_kernel(liso_targets_cfg c, const float* __restrict__ box_pos,
                                                              const float* __restrict__ box_dims, const float* __restrict__ box_rot,
                                                              float* __restrict__ box_max) {
    const int bk = blockIdx.x;  // one block per (sample, box)
    const float bx = box_pos[3 * bk], by = box_pos[3 * bk + 1], len = box_dims[3 * bk], wid = box_dims[3 * bk + 1];
    const float cs = cosf(box_rot[bk]), sn = sinf(box_rot[bk]);
    float m = 0.f;
    for (int i = threadIdx.x; i < c.h * c.w; i += 256) {
        const int y = i / c.w, x = i - y * c.w;
        m = fmaxf(m, box_heat(cell_center(y, c.h, c.range_x), cell_center(x, c.w, c.range_y), bx, by, cs, sn, len, wid));
    }
    __shared__ float red[4];
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) box_max[bk] = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-5f);  // :111-115
}

__global__ __launch_bounds__(256) void targets_render_kernel(liso_targets_cfg c, const float* __restrict__ box_pos,
                                                             const float* __restrict__ box_dims, const float* __restrict__ box_rot,
                                                             const uint8_t* __restrict__ box_valid, const float* __restrict__ box_max,
                                                             float* __restrict__ probs, float* __restrict__ dims,
                                                             float* __restrict__ pos, float* __restrict__ rot,
                                                             uint8_t* __restrict__ center_mask) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long n = (long)c.batch * c.h * c.w;
    if (i >= n) return;
    const int b = (int)(i / ((long)c.h * c.w));
    const int rem = (int)(i - (long)b * c.h * c.w);
    const int y = rem / c.w, x = rem - y * c.w;
    const float cx = cell_center(y, c.h, c.range_x), cy = cell_center(x, c.w, c.range_y);
    const float* P = box_pos + (size_t)b * c.n_boxes * 3;
    const float* D = box_dims + (size_t)b * c.n_boxes * 3;
    const float* R = box_rot + (size_t)b * c.n_boxes;
    const uint8_t* V = box_valid + (size_t)b * c.n_boxes;
    const float* M = box_max + (size_t)b * c.n_boxes;
    float best = 0.f;
    int center = 0;
    for (int k = 0; k < c.n_boxes; k++) {
        if (!V[k]) continue;
        const float h = box_heat(cx, cy, P[3 * k], P[3 * k + 1], cosf(R[k]), sinf(R[k]), D[3 * k], D[3 * k + 1]) / M[k];
        best = fmaxf(best, h);
        // the cell that contains the box centre (create_occupancy_pcl_image of the centres, :309-314)
        const int iy = (int)((P[3 * k] + 0.5f * c.range_x) * ((float)c.h / c.range_x));
        const int ix = (int)((P[3 * k + 1] + 0.5f * c.range_y) * ((float)c.w / c.range_y));
        center |= (iy == y && ix == x && P[3 * k] + 0.5f * c.range_x >= 0.f && P[3 * k + 1] + 0.5f * c.range_y >= 0.f);
    }
    float ad[3] = {0.f, 0.f, 0.f}, ap[3] = {0.f, 0.f, 0.f}, ar[2] = {0.f, 0.f};
    if (best > 0.01f) {  // occupancy threshold, :212-215; the hottest box(es) of the cell give it their attributes
        for (int k = 0; k < c.n_boxes; k++) {
            if (!V[k]) continue;
            const float h = box_heat(cx, cy, P[3 * k], P[3 * k + 1], cosf(R[k]), sinf(R[k]), D[3 * k], D[3 * k + 1]) / M[k];
            if (h == best) {
#pragma unroll
                for (int j = 0; j < 3; j++) { ad[j] += D[3 * k + j]; ap[j] += P[3 * k + j]; }
                ar[0] += sinf(R[k]); ar[1] += cosf(R[k]);
            }
        }
    }
    probs[i] = best;
#pragma unroll
    for (int j = 0; j < 3; j++) { dims[3 * i + j] = ad[j]; pos[3 * i + j] = ap[j]; }
    rot[2 * i] = ar[0]; rot[2 * i + 1] = ar[1];
    center_mask[i] = (uint8_t)center;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }
inline int n_blocks(const liso_centerloss_cfg* c) {
    const long n = (long)c->batch * c->h * c->w;
    const long b = (n + kThreads - 1) / kThreads;
    return (int)(b < 512 ? (b > 0 ? b : 1) : 512);
}
inline bool cfg_ok(const liso_centerloss_cfg* c) { return c && c->batch >= 1 && c->h >= 1 && c->w >= 1; }
inline Maps make_maps(const float* pos, const float* dims, const float* rot, const float* probs, const long* strides) {
    Maps m;
    m.pos = pos; m.dims = dims; m.rot = rot; m.probs = probs;
    for (int k = 0; k < 16; k++) m.s[k] = strides[k];
    return m;
}

}  // namespace

extern "C" {

size_t liso_centerloss_workspace_bytes(const liso_centerloss_cfg* cfg) {
    return cfg_ok(cfg) ? (size_t)n_blocks(cfg) * NS * sizeof(double) : 0;
}

int liso_centerloss_fwd_f32(const liso_centerloss_cfg* cfg, const float* pos, const float* dims, const float* rot,
                            const float* probs, const long* strides, const float* gt_probs, const float* gt_dims,
                            const float* gt_pos, const float* gt_rot, const uint8_t* center_mask, const uint8_t* ignore_mask,
                            const float* rot_weights, const float* pillar_centers, double* sums, float* losses,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!cfg_ok(cfg) || !pos || !dims || !rot || !probs || !strides || !gt_probs || !gt_dims || !gt_pos || !gt_rot ||
        !center_mask || !pillar_centers || !sums || !losses || !workspace)
        return LISO_EINVAL;
    if (workspace_bytes < liso_centerloss_workspace_bytes(cfg)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = n_blocks(cfg);
    centerloss_partial_kernel<<<nb, kThreads, 0, st>>>(*cfg, make_maps(pos, dims, rot, probs, strides), gt_probs, gt_dims, gt_pos,
                                                       gt_rot, center_mask, ignore_mask, rot_weights, pillar_centers,
                                                       (double*)workspace);
    centerloss_final_kernel<<<1, NS * 64, 0, st>>>(*cfg, (const double*)workspace, nb, sums, losses);
    return check_launch();
}

int liso_centerloss_bwd_f32(const liso_centerloss_cfg* cfg, const float* pos, const float* dims, const float* rot,
                            const float* probs, const long* strides, const float* gt_probs, const float* gt_dims,
                            const float* gt_pos, const float* gt_rot, const uint8_t* center_mask, const uint8_t* ignore_mask,
                            const float* rot_weights, const float* pillar_centers, const double* sums,
                            const float* grad_total, float* g_pos, float* g_dims, float* g_rot, float* g_probs, void* stream) {
    if (!cfg_ok(cfg) || !pos || !dims || !rot || !probs || !strides || !gt_probs || !gt_dims || !gt_pos || !gt_rot ||
        !center_mask || !pillar_centers || !sums || !grad_total || !g_pos || !g_dims || !g_rot || !g_probs)
        return LISO_EINVAL;
    const long n = (long)cfg->batch * cfg->h * cfg->w;
    centerloss_bwd_kernel<<<(unsigned)((n + kThreads - 1) / kThreads), kThreads, 0, (hipStream_t)stream>>>(
        *cfg, make_maps(pos, dims, rot, probs, strides), gt_probs, gt_dims, gt_pos, gt_rot, center_mask, ignore_mask, rot_weights,
        pillar_centers, sums, grad_total, g_pos, g_dims, g_rot, g_probs);
    return check_launch();
}

int liso_render_center_targets_f32(const liso_targets_cfg* cfg, const float* box_pos, const float* box_dims, const float* box_rot,
                                   const uint8_t* box_valid, float* box_max, float* probs, float* dims, float* pos, float* rot,
                                   uint8_t* center_mask, void* stream) {
    if (!cfg || cfg->batch < 1 || cfg->n_boxes < 0 || cfg->h < 1 || cfg->w < 1 || !probs || !dims || !pos || !rot || !center_mask)
        return LISO_EINVAL;
    if (cfg->n_boxes > 0 && (!box_pos || !box_dims || !box_rot || !box_valid || !box_max)) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (cfg->n_boxes > 0)
        targets_box_max_kernel<<<cfg->batch * cfg->n_boxes, 256, 0, st>>>(*cfg, box_pos, box_dims, box_rot, box_max);
    const long n = (long)cfg->batch * cfg->h * cfg->w;
    targets_render_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(*cfg, box_pos, box_dims, box_rot, box_valid, box_max, probs,
                                                                      dims, pos, rot, center_mask);
    return check_launch();
}

}  // extern "C"
