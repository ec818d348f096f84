// Flow -> pseudo-box helper stages for gfx950 (MI355X).  C ABI + reference lines: include/liso_flow_cluster.h.
//
//   bev_scatter_kernel   one thread per point: nonrigid flow in fp64, then 4 order-independent 64-bit fixed-point
//                        atomics + 1 count atomic into the pillar grid (the cloud is read once: 12+12+8+1 B/point)
//   bev_mean_kernel      one thread per pillar: fixed point -> fp32, divide where count > 1
//   fit_z_kernel         lanes = boxes, every lane walks the point tile from LDS (broadcast reads) keeping count /
//                        min / max of the in-box z in registers: no [N,K,4] fp64 tensor (192 MB at N=120k, K=50)
//   fit_z_final_kernel   fixed-order combine of the block partials (min/max/sum are order independent)
#include <hip/hip_runtime.h>
#include "zero_fill.h"
#include <math.h>
#include <stdint.h>

#include "../../include/liso_flow_cluster.h"
#include "../../include/liso_iou3d.h"

namespace {

constexpr double kFix = 16777216.0;  // 2^24 fixed-point scale: 6e-8 m resolution, |sum| < 5e11 m fits int64

__global__ __launch_bounds__(256) void bev_scatter_kernel(const float* __restrict__ points, int ps, const uint8_t* __restrict__ valid,
                                   const int32_t* __restrict__ coors, const float* __restrict__ flow, int fs,
                                   const double* __restrict__ ome, int batch, int n, int h, int w,
                                   long long* __restrict__ sums, int* __restrict__ counts) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    long long cell = -1;
    long long q[4] = {0, 0, 0, 0};
    if (i < (size_t)batch * n && valid[i]) {  // masked_scatter_mean_2d only scatters valid rows (:63-66)
        const int b = (int)(i / n);
        const int r = coors[i * 2 + 0], c = coors[i * 2 + 1];
        if (r >= 0 && r < h && c >= 0 && c < w) {
            const float* p = points + i * ps;
            const float* f = flow + i * fs;
            const double* M = ome + (size_t)b * 16;
            const double x = p[0], y = p[1], z = p[2];
            float nr[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                // bev_flow_utils.py:28-41: static flow = ((inv(odom) - I) [x y z 1])[:3] in fp64, cast to fp32, subtracted in fp32
                const float stat = (float)(M[k * 4 + 0] * x + M[k * 4 + 1] * y + M[k * 4 + 2] * z + M[k * 4 + 3]);
                nr[k] = f[k] - stat;
            }
            // torch.linalg.norm (fp32)
            const float len = sqrtf(nr[0] * nr[0] + nr[1] * nr[1] + nr[2] * nr[2]);
            cell = (long long)(((size_t)b * h + r) * w + c);
            q[0] = llrint((double)len * kFix);
            q[1] = llrint((double)nr[0] * kFix);
            q[2] = llrint((double)nr[1] * kFix);
            q[3] = llrint((double)nr[2] * kFix);
        }
    }
    // consecutive returns of one ring fall into the same pillar: the lanes of a RUN of equal cells add up in the wave (integer sums:
    // any grouping gives the same bits) and the run's first lane issues the five atomics for all of them
    const long long prev = __shfl_up(cell, 1);
    const bool head = lane == 0 || prev != cell;
    const unsigned long long heads = __ballot(head);
    const int run = __popcll(heads & ((2ull << lane) - 1ull));  // run index of this lane (1-based)
    int cnt = cell >= 0 ? 1 : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int other_run = __shfl_down(run, o);
        const bool take = lane + o < 64 && other_run == run;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long v = __shfl_down(q[k], o);
            if (take) q[k] += v;
        }
        const int vc = __shfl_down(cnt, o);
        if (take) cnt += vc;
    }
    if (!head || cell < 0) return;
    unsigned long long* s = (unsigned long long*)(sums + cell * 4);
    atomicAdd(s + 0, (unsigned long long)q[0]);
    atomicAdd(s + 1, (unsigned long long)q[1]);
    atomicAdd(s + 2, (unsigned long long)q[2]);
    atomicAdd(s + 3, (unsigned long long)q[3]);
    atomicAdd(&counts[cell], cnt);
}

__global__ void bev_mean_kernel(const long long* __restrict__ sums, const int* __restrict__ counts, size_t cells,
                                float* __restrict__ dyn, float* __restrict__ nrflow) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    const int cnt = counts[i];
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        v[k] = (float)((double)sums[i * 4 + k] / kFix);
        if (cnt > 1) v[k] = v[k] / (float)cnt;  // torch_differentiable_forward_scatter.py:84-86
    }
    dyn[i] = v[0];
    nrflow[i * 3 + 0] = v[1]; nrflow[i * 3 + 1] = v[2]; nrflow[i * 3 + 2] = v[3];
}

constexpr int kFitThreads = 256;
constexpr int kFitTile = 256;
constexpr int kFitMaxBlocks = 256;

struct FitPartial {
    int count;
    float zmin, zmax, zmin_sensor;
    int zmin_idx;
};

// flow_cluster_detector.py:339-384
__global__ __launch_bounds__(kFitThreads) void fit_z_kernel(const float* __restrict__ points, int ps, int n,
                                                            const float* __restrict__ box_pos, int pos_dims,
                                                            const float* __restrict__ box_dims, int dims_dims,
                                                            const float* __restrict__ box_rot, int k, float box_height,
                                                            FitPartial* __restrict__ partials, int tiles_per_block) {
    __shared__ float px[kFitTile], py[kFitTile], pz[kFitTile];
    __shared__ FitPartial red[kFitThreads / 64][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int box = blockIdx.y * 64 + lane;
    const bool has = box < k;
    double bx = 0, by = 0, bz = 0, c = 1, s = 0;
    float hx = 0, hy = 0, hz = 0;
    if (has) {
        bx = box_pos[box * pos_dims + 0]; by = box_pos[box * pos_dims + 1];
        bz = pos_dims == 3 ? (double)box_pos[box * pos_dims + 2] : 0.0;  // shape_utils.py:280: t_z None -> 0
        const double th = (double)box_rot[box];
        c = cos(th); s = sin(th);
        hx = 0.5f * box_dims[box * dims_dims + 0]; hy = 0.5f * box_dims[box * dims_dims + 1];
        hz = 0.5f * (dims_dims == 3 ? box_dims[box * dims_dims + 2] : box_height);
    }
    FitPartial acc = {0, box_height, -box_height, 0.f, 0x7fffffff};
    const int tile0 = blockIdx.x * tiles_per_block;
    for (int t = 0; t < tiles_per_block; t++) {
        const int base = (tile0 + t) * kFitTile;
        if (base >= n) break;
        {
            const int i = base + tid;
            if (i < n) { px[tid] = points[(size_t)i * ps]; py[tid] = points[(size_t)i * ps + 1]; pz[tid] = points[(size_t)i * ps + 2]; }
        }
        __syncthreads();
        const int cntp = n - base < kFitTile ? n - base : kFitTile;
        if (has) {
            // every wave walks a quarter of the tile
            for (int j = wave; j < cntp; j += kFitThreads / 64) {
                const double dx = (double)px[j] - bx, dy = (double)py[j] - by;
                // inverse pose applied in fp64, result cast to fp32 before the comparison (:354-360)
                const float lx = (float)(c * dx + s * dy), ly = (float)(-s * dx + c * dy), lz = (float)((double)pz[j] - bz);
                if (fabsf(lx) < hx && fabsf(ly) < hy && fabsf(lz) < hz) {
                    acc.count++;
                    acc.zmax = fmaxf(acc.zmax, lz);
                    const int gi = base + j;
                    if (lz < acc.zmin || (lz == acc.zmin && gi < acc.zmin_idx)) { acc.zmin = lz; acc.zmin_sensor = pz[j]; acc.zmin_idx = gi; }
                }
            }
        }
        __syncthreads();
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && has) {
        FitPartial r = red[0][lane];
        for (int wv = 1; wv < kFitThreads / 64; wv++) {
            const FitPartial o = red[wv][lane];
            r.count += o.count;
            r.zmax = fmaxf(r.zmax, o.zmax);
            if (o.zmin < r.zmin || (o.zmin == r.zmin && o.zmin_idx < r.zmin_idx)) { r.zmin = o.zmin; r.zmin_sensor = o.zmin_sensor; r.zmin_idx = o.zmin_idx; }
        }
        partials[(size_t)blockIdx.x * k + box] = r;
    }
}

// one wavefront per box: the lanes stride over the block partials (up to 1024 of them: a single thread walking them took 78 us),
// then a butterfly combines count (sum), zmax (max) and the lowest point (zmin, then smallest index) -- all order independent
__global__ __launch_bounds__(64) void fit_z_final_kernel(const FitPartial* __restrict__ partials, int nblk, int k,
                                                         const float* __restrict__ points, int ps, int n, float box_height,
                                                         long long* __restrict__ num_pts, float* __restrict__ fz,
                                                         float* __restrict__ fh) {
    const int box = blockIdx.x, lane = threadIdx.x;
    if (box >= k) return;
    FitPartial r = {0, box_height, -box_height, 0.f, 0x7fffffff};
    for (int b = lane; b < nblk; b += 64) {
        const FitPartial o = partials[(size_t)b * k + box];
        r.count += o.count;
        r.zmax = fmaxf(r.zmax, o.zmax);
        if (o.zmin < r.zmin || (o.zmin == r.zmin && o.zmin_idx < r.zmin_idx)) { r.zmin = o.zmin; r.zmin_sensor = o.zmin_sensor; r.zmin_idx = o.zmin_idx; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        FitPartial o;
        o.count = __shfl_xor(r.count, off);
        o.zmin = __shfl_xor(r.zmin, off);
        o.zmax = __shfl_xor(r.zmax, off);
        o.zmin_sensor = __shfl_xor(r.zmin_sensor, off);
        o.zmin_idx = __shfl_xor(r.zmin_idx, off);
        r.count += o.count;
        r.zmax = fmaxf(r.zmax, o.zmax);
        if (o.zmin < r.zmin || (o.zmin == r.zmin && o.zmin_idx < r.zmin_idx)) { r.zmin = o.zmin; r.zmin_sensor = o.zmin_sensor; r.zmin_idx = o.zmin_idx; }
    }
    if (lane != 0) return;
    // :367-372 height = clip(zmax - zmin, 1, 2); z = sensor z of the lowest in-box point + h/2.  An empty box has
    // zmin == +box_height for every point, argmin == 0: the reference then takes point 0's z.
    const float height = fminf(fmaxf(r.zmax - r.zmin, 1.0f), 2.0f);
    const float zlow = r.count > 0 ? r.zmin_sensor : (n > 0 ? points[2] : 0.f);
    num_pts[box] = r.count;
    fh[box] = height;
    fz[box] = zlow + 0.5f * height;
}

// inv(M) - I of B row-major 4x4 fp64 matrices by cofactor expansion (bev_flow_utils.py:30-33: torch.linalg.inv(odom) - eye): one
// thread per matrix.  A singular matrix gives inf / nan entries like the reference's LU inverse.
__global__ void odom_inverse_minus_eye_kernel(const double* __restrict__ m_all, int batch, double* __restrict__ out_all) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double* m = m_all + (size_t)b * 16;
    double* o = out_all + (size_t)b * 16;
    const double a00 = m[0], a01 = m[1], a02 = m[2], a03 = m[3], a10 = m[4], a11 = m[5], a12 = m[6], a13 = m[7];
    const double a20 = m[8], a21 = m[9], a22 = m[10], a23 = m[11], a30 = m[12], a31 = m[13], a32 = m[14], a33 = m[15];
    const double s0 = a00 * a11 - a10 * a01, s1 = a00 * a12 - a10 * a02, s2 = a00 * a13 - a10 * a03;
    const double s3 = a01 * a12 - a11 * a02, s4 = a01 * a13 - a11 * a03, s5 = a02 * a13 - a12 * a03;
    const double c5 = a22 * a33 - a32 * a23, c4 = a21 * a33 - a31 * a23, c3 = a21 * a32 - a31 * a22;
    const double c2 = a20 * a33 - a30 * a23, c1 = a20 * a32 - a30 * a22, c0 = a20 * a31 - a30 * a21;
    const double inv = 1.0 / (s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0);
    o[0] = (a11 * c5 - a12 * c4 + a13 * c3) * inv - 1.0;
    o[1] = (-a01 * c5 + a02 * c4 - a03 * c3) * inv;
    o[2] = (a31 * s5 - a32 * s4 + a33 * s3) * inv;
    o[3] = (-a21 * s5 + a22 * s4 - a23 * s3) * inv;
    o[4] = (-a10 * c5 + a12 * c2 - a13 * c1) * inv;
    o[5] = (a00 * c5 - a02 * c2 + a03 * c1) * inv - 1.0;
    o[6] = (-a30 * s5 + a32 * s2 - a33 * s1) * inv;
    o[7] = (a20 * s5 - a22 * s2 + a23 * s1) * inv;
    o[8] = (a10 * c4 - a11 * c2 + a13 * c0) * inv;
    o[9] = (-a00 * c4 + a01 * c2 - a03 * c0) * inv;
    o[10] = (a30 * s4 - a31 * s2 + a33 * s0) * inv - 1.0;
    o[11] = (-a20 * s4 + a21 * s2 - a23 * s0) * inv;
    o[12] = (-a10 * c3 + a11 * c1 - a12 * c0) * inv;
    o[13] = (a00 * c3 - a01 * c1 + a02 * c0) * inv;
    o[14] = (-a30 * s3 + a31 * s1 - a32 * s0) * inv;
    o[15] = (a20 * s3 - a21 * s1 + a22 * s0) * inv - 1.0;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

inline int fit_blocks(int n) {
    const int tiles = (n + kFitTile - 1) / kFitTile;
    return tiles < kFitMaxBlocks ? (tiles > 0 ? tiles : 1) : kFitMaxBlocks;
}

}  // namespace

extern "C" {

size_t liso_bev_dynamic_flow_workspace_bytes(int batch, int h, int w) {
    if (batch <= 0 || h <= 0 || w <= 0) return 0;
    return (size_t)batch * h * w * (4 * sizeof(long long) + sizeof(int));
}

int liso_bev_dynamic_flow_f32(const float* points, int point_stride, const uint8_t* valid, const int32_t* pillar_coors,
                              const float* flow, int flow_stride, const double* odom_minus_eye, int batch, int n, int h,
                              int w, float* dynamicness, float* nonrigid_flow, void* workspace, size_t workspace_bytes,
                              void* stream) {
    if (batch <= 0 || n < 0 || h <= 0 || w <= 0 || point_stride < 3 || flow_stride < 3) return LISO_EINVAL;
    if (!dynamicness || !nonrigid_flow || !workspace || !odom_minus_eye) return LISO_EINVAL;
    if (n > 0 && (!points || !valid || !pillar_coors || !flow)) return LISO_EINVAL;
    const size_t need = liso_bev_dynamic_flow_workspace_bytes(batch, h, w);
    if (workspace_bytes < need) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)batch * h * w;
    long long* sums = (long long*)workspace;
    int* counts = (int*)(sums + cells * 4);
    if (liso_zero::zero_async(workspace, need, st) != hipSuccess) return LISO_ELAUNCH;
    const size_t total = (size_t)batch * n;
    if (total > 0)
        bev_scatter_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(points, point_stride, valid, pillar_coors, flow,
                                                                          flow_stride, odom_minus_eye, batch, n, h, w, sums,
                                                                          counts);
    bev_mean_kernel<<<(unsigned)((cells + 255) / 256), 256, 0, st>>>(sums, counts, cells, dynamicness, nonrigid_flow);
    return check_launch();
}

int liso_odom_inverse_minus_eye_f64(const double* odom, int batch, double* out, void* stream) {
    if (batch < 0 || (batch > 0 && (!odom || !out))) return LISO_EINVAL;
    if (batch == 0) return LISO_OK;
    odom_inverse_minus_eye_kernel<<<(batch + 63) / 64, 64, 0, (hipStream_t)stream>>>(odom, batch, out);
    return check_launch();
}

size_t liso_fit_box_z_workspace_bytes(int n_points, int n_boxes) {
    if (n_points < 0 || n_boxes <= 0) return 0;
    return (size_t)fit_blocks(n_points) * n_boxes * sizeof(FitPartial);
}

int liso_fit_box_z_f32(const float* points, int point_stride, int n, const float* box_pos, int pos_dims,
                       const float* box_dims, int dims_dims, const float* box_rot, int k, float box_height,
                       int64_t* num_pts, float* fitted_z, float* fitted_height, void* workspace, size_t workspace_bytes,
                       void* stream) {
    if (n < 0 || k < 0 || point_stride < 3 || (pos_dims != 2 && pos_dims != 3) || (dims_dims != 2 && dims_dims != 3))
        return LISO_EINVAL;
    if (k == 0) return LISO_OK;
    if (!box_pos || !box_dims || !box_rot || !num_pts || !fitted_z || !fitted_height || !workspace || (n > 0 && !points))
        return LISO_EINVAL;
    if (workspace_bytes < liso_fit_box_z_workspace_bytes(n, k)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = fit_blocks(n);
    const int tiles = (n + kFitTile - 1) / kFitTile;
    const int tpb = (tiles + nblk - 1) / nblk > 0 ? (tiles + nblk - 1) / nblk : 1;
    fit_z_kernel<<<dim3(nblk, (k + 63) / 64), kFitThreads, 0, st>>>(points, point_stride, n, box_pos, pos_dims, box_dims,
                                                                    dims_dims, box_rot, k, box_height,
                                                                    (FitPartial*)workspace, tpb);
    fit_z_final_kernel<<<k, 64, 0, st>>>((const FitPartial*)workspace, nblk, k, points, point_stride, n,
                                                     box_height, (long long*)num_pts, fitted_z, fitted_height);
    return check_launch();
}

}  // extern "C"
