// Points-in-boxes statistics (count, mean flow, optional mask) and greedy detection<->ground-truth matching for gfx950.
// C ABI + reference lines: include/liso_tracking.h.
//
// points_in_boxes: one block = 256 consecutive points of one batch row x one tile of 128 boxes held in LDS (inverse pose +
// half extents, 64 B per box, read as LDS broadcasts: every lane of a wavefront tests the same box against its own point).
// A fp32 circle test rejects most (point, box) pairs before the exact transform.  Hits are rare (a point lies in <= a few
// boxes), so the per-box accumulators live in LDS and take integer atomics; a block adds its non-zero accumulators to HBM
// once.  120k points x 100 boxes = 469 blocks, x 1000 boxes = 3752 blocks (>> 256 CUs).
#include <hip/hip_runtime.h>
#include "zero_fill.h"
#include <math.h>
#include <stdint.h>

#include <cstring>
#include <rocprim/warp/warp_reduce.hpp>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_tracking.h"

namespace {

constexpr int kThreads = 256;  // one point per thread
constexpr int kTile = 128;     // boxes per block (LDS tile)
constexpr double kFixedScale = 16777216.0;  // 2^24 per metre

// One box in LDS, 64 B: rows x and y of inv(sensor_T_box) = [Rz(yaw)^T | -Rz^T pos] are (c, s, 0, m03) and (-s, c, 0, m13),
// row z is (0, 0, 1, tz); the zero entries are not stored (adding 0 * z changes nothing for finite z; non-finite points
// are excluded before the test).
struct BoxRow {
    double c, s, m03, m13, tz;
    float hx, hy, hz;  // 0.5 * bloat * dims
    float pad;
};
struct PreRow {
    float x, y, r2;  // conservative circle around the box footprint: fp32 reject before the exact test
    int count;
};

template <int PREC>
__device__ __forceinline__ bool inside(const BoxRow& r, float px, float py, float pz) {
    float bx, by, bz;
    if (PREC == 0) {  // fp64 product, rounded to fp32 (torch_dataset_commons.py:1914-1918)
        const double dx = px, dy = py, dz = pz;
        bx = (float)(r.c * dx + r.s * dy + r.m03);
        by = (float)(r.c * dy - r.s * dx + r.m13);
        bz = (float)(dz + r.tz);
    } else {          // inverse rounded to fp32, fp32 product (shape_utils.py:514-518)
        const float c = (float)r.c, s = (float)r.s;
        bx = fmaf(s, py, c * px) + (float)r.m03;
        by = fmaf(c, py, -s * px) + (float)r.m13;
        bz = pz + (float)r.tz;
    }
    return fabsf(bx) < r.hx && fabsf(by) < r.hy && fabsf(bz) < r.hz;
}

template <int PREC>
__global__ __launch_bounds__(kThreads) void points_in_boxes_kernel(liso_boxpts_cfg c, const float* __restrict__ boxes,
                                                                   const float* __restrict__ points,
                                                                   const uint8_t* __restrict__ point_valid,
                                                                   const float* __restrict__ flow, uint8_t* __restrict__ mask,
                                                                   int* __restrict__ count, long long* __restrict__ fsum) {
    __shared__ BoxRow rows[kTile];
    __shared__ PreRow pre[kTile];
    __shared__ long long fs[kTile][3];
    const int b = blockIdx.z, tile0 = blockIdx.y * kTile, tk = min(kTile, c.k - tile0);
    if ((int)threadIdx.x < tk) {
        const int j = threadIdx.x;
        const float* box = boxes + ((size_t)b * c.k + tile0 + j) * 7;
        // Shape.get_poses: sensor_T_box = [Rz(yaw) | pos] in fp64 (shape_utils.py:271-319)
        const double x = box[0], y = box[1], z = box[2], yaw = box[6];
        const double cs = cos(yaw), sn = sin(yaw);
        BoxRow r;
        r.c = cs; r.s = sn; r.m03 = -(cs * x + sn * y); r.m13 = sn * x - cs * y; r.tz = -z;
        r.hx = 0.5f * (c.dims_bloat * box[3]); r.hy = 0.5f * (c.dims_bloat * box[4]); r.hz = 0.5f * (c.dims_bloat * box[5]);
        r.pad = 0.f;
        rows[j] = r;
        // inside => bx^2 + by^2 < hx^2 + hy^2; the margin covers the fp32 rounding of the squared distance at |xy| <= 1e4 m
        const float r2 = r.hx * r.hx + r.hy * r.hy;
        pre[j].x = box[0]; pre[j].y = box[1];
        pre[j].r2 = r2 * 1.001f + 0.05f;  // NaN boxes: every comparison below is false -> never inside
        pre[j].count = 0;
        fs[j][0] = fs[j][1] = fs[j][2] = 0;
    }
    __syncthreads();
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;  // consecutive lanes read consecutive rows
    if (i < c.n) {
        const float* p = points + ((size_t)b * c.n + i) * c.point_stride;
        const float px = p[0], py = p[1], pz = p[2];
        const bool finite = isfinite(px) && isfinite(py) && isfinite(pz);
        long long fx[3] = {0, 0, 0};
        if (fsum != nullptr && finite && (point_valid == nullptr || point_valid[(size_t)b * c.n + i])) {
            const float* f = flow + ((size_t)b * c.n + i) * 3;
            for (int a = 0; a < 3; ++a) fx[a] = isfinite(f[a]) ? (long long)llrint((double)f[a] * kFixedScale) : 0;
        }
        uint8_t* mrow = mask != nullptr ? mask + ((size_t)b * c.n + i) * c.k + tile0 : nullptr;
        const bool pack = mrow != nullptr && (c.k % 4 == 0);  // 4 boxes per 32-bit store (tile0 % 4 == 0)
        uint32_t word = 0;
        for (int j = 0; j < tk; ++j) {
            const float ex = px - pre[j].x, ey = py - pre[j].y;
            bool in = false;
            if (finite && ex * ex + ey * ey < pre[j].r2) in = inside<PREC>(rows[j], px, py, pz);
            if (in) {
                atomicAdd(&pre[j].count, 1);
                for (int a = 0; a < 3; ++a)
                    if (fx[a] != 0) atomicAdd((unsigned long long*)&fs[j][a], (unsigned long long)fx[a]);
            }
            if (pack) {
                word |= (uint32_t)in << (8 * (j & 3));
                if ((j & 3) == 3) {
                    *(uint32_t*)(mrow + j - 3) = word;
                    word = 0;
                }
            } else if (mrow != nullptr) {
                mrow[j] = in;
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < tk && pre[threadIdx.x].count != 0) {  // one HBM atomic per box that was hit in this block
        const size_t o = (size_t)b * c.k + tile0 + threadIdx.x;
        if (count != nullptr) atomicAdd(count + o, pre[threadIdx.x].count);
        if (fsum != nullptr)
            for (int a = 0; a < 3; ++a)
                if (fs[threadIdx.x][a] != 0)
                    atomicAdd((unsigned long long*)(fsum + 3 * o + a), (unsigned long long)fs[threadIdx.x][a]);
    }
}

__global__ void mean_flow_kernel(long rows, const int* __restrict__ count, const long long* __restrict__ fsum,
                                 float* __restrict__ mean_flow) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float denom = fmaxf((float)count[r], 1.f);  // torch.clip(point_is_in_box.sum(dim=1), min=1.0) (tracking.py:2185)
    for (int a = 0; a < 3; ++a) mean_flow[3 * r + a] = (float)((double)fsum[3 * r + a] / kFixedScale) / denom;
}

// One wavefront walks the predictions in confidence order (box_groundtruth_matching_iou.py:33-68).  Lane l owns the
// ground-truth rows l, l+64, ...  REGS > 0 (n_gt <= 64 * REGS): those rows' "taken" flags live in a per-lane bit mask, the
// visiting order sits in LDS and the IoU columns of the next kDepth predictions are in flight while the current kDepth are
// reduced (the walk is a chain of dependent L2 round trips otherwise).  REGS == 0: any n_gt, flags in HBM.
constexpr int kDepth = 4;
constexpr int kLdsOrder = 8192;

template <int REGS>
__global__ __launch_bounds__(64) void greedy_match_kernel(const float* __restrict__ iou, long gs, long ps, int n_gt, int n_pred,
                                                          const long long* __restrict__ order, float thr,
                                                          long long* __restrict__ idx_gt, long long* __restrict__ idx_pred,
                                                          float* __restrict__ match_iou, int* __restrict__ num_matches,
                                                          uint8_t* __restrict__ pred_mask, uint8_t* __restrict__ gt_mask) {
    constexpr int R = REGS > 0 ? REGS : 1;
    __shared__ int sorder[kLdsOrder];
    __shared__ typename rocprim::warp_reduce<unsigned long long, 64, true>::storage_type reduce_storage;
    const int lane = threadIdx.x;
    const bool lds_order = n_pred <= kLdsOrder;
    for (int i = lane; i < n_pred; i += 64) {
        pred_mask[i] = 0;
        const long long p = order[i];
        if (lds_order) sorder[i] = (p >= 0 && p < n_pred) ? (int)p : -1;
    }
    for (int i = lane; i < n_gt; i += 64) gt_mask[i] = 0;
    __syncthreads();
    auto pred_at = [&](int t) -> int {
        if (t >= n_pred) return -1;
        if (lds_order) return sorder[t];
        const long long p = order[t];
        return (p >= 0 && p < n_pred) ? (int)p : -1;
    };
    int m = 0;
    uint32_t taken = 0;
    float cur[kDepth][R], nxt[kDepth][R];
    auto load_group = [&](int t0) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int p = pred_at(t0 + d);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int g = lane + 64 * r;
                nxt[d][r] = (p >= 0 && g < n_gt) ? iou[g * gs + p * ps] : nanf("");  // NaN is never chosen
            }
        }
    };
    if (REGS > 0) load_group(0);
    for (int t0 = 0; t0 < n_pred; t0 += kDepth) {
        if (REGS > 0) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d)
#pragma unroll
                for (int r = 0; r < R; ++r) cur[d][r] = nxt[d][r];
            load_group(t0 + kDepth);  // in flight during the reductions below
        }
#pragma unroll
        for (int d = 0; d < kDepth; ++d) {
            const int p = pred_at(t0 + d);
            if (t0 + d >= n_pred) break;
            // key = (order-preserving bits of the IoU) << 32 | ~g: the maximum key is the largest IoU and, among equal IoUs,
            // the smallest ground-truth index (the reference's strict > over ascending gt_idx); 0 = no candidate
            unsigned long long key = 0;
            auto candidate = [&](float v, int g) {
                if (!(v > -INFINITY)) return;  // NaN and -inf never beat the initial max_iou = -inf
                const uint32_t u = __float_as_uint(v);
                const unsigned long long k = ((unsigned long long)(u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u)) << 32) | (0xFFFFFFFFu - (uint32_t)g);
                key = k > key ? k : key;
            };
            if (REGS > 0) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (!((taken >> r) & 1u)) candidate(cur[d][r], lane + 64 * r);
            } else if (p >= 0) {
                for (int g = lane; g < n_gt; g += 64)
                    if (!gt_mask[g]) candidate(iou[g * gs + p * ps], g);
            }
            unsigned long long top;
            rocprim::warp_reduce<unsigned long long, 64, true>().reduce(key, top, reduce_storage, rocprim::maximum<unsigned long long>());
            const uint32_t su = (uint32_t)(top >> 32);
            const float best = __uint_as_float(su ^ ((su >> 31) ? 0x80000000u : 0xFFFFFFFFu));
            const int bi = top != 0 ? (int)(0xFFFFFFFFu - (uint32_t)top) : INT32_MAX;
            if (bi != INT32_MAX && best > thr) {  // every lane holds the same (best, bi)
                if (REGS > 0 && (bi & 63) == lane) taken |= 1u << (bi >> 6);
                if (lane == 0) {
                    idx_gt[m] = bi;
                    idx_pred[m] = p;
                    match_iou[m] = best;
                    gt_mask[bi] = 1;
                    pred_mask[p] = 1;
                }
                ++m;
            }
            if (REGS == 0) __syncthreads();  // the taken flag in HBM is visible to the next prediction's scan
        }
    }
    if (lane == 0) *num_matches = m;
}

}  // namespace

extern "C" size_t liso_points_in_boxes_workspace_bytes(const liso_boxpts_cfg* c) {
    if (c == nullptr || c->batch <= 0 || c->k <= 0) return 0;
    return (size_t)c->batch * c->k * 3 * sizeof(long long);
}

extern "C" int liso_points_in_boxes_f32(const liso_boxpts_cfg* c, const float* boxes, const float* points,
                                        const uint8_t* point_valid, const float* flow, uint8_t* mask, int* count,
                                        float* mean_flow, void* workspace, size_t workspace_bytes, void* stream) {
    if (c == nullptr || c->batch < 0 || c->n < 0 || c->k < 0 || c->point_stride < 3 || (c->precision != 0 && c->precision != 1))
        return LISO_EINVAL;
    if (mean_flow != nullptr && ((flow == nullptr && c->n > 0) || count == nullptr)) return LISO_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)c->batch * c->k;
    if (rows == 0) return LISO_OK;
    if (boxes == nullptr || (c->n > 0 && points == nullptr)) return LISO_EINVAL;
    long long* fsum = nullptr;
    if (mean_flow != nullptr) {
        if (workspace == nullptr || workspace_bytes < liso_points_in_boxes_workspace_bytes(c)) return LISO_EWORKSPACE;
        fsum = (long long*)workspace;
        if (liso_zero::zero_async(fsum, rows * 3 * sizeof(long long), s) != hipSuccess) return LISO_ELAUNCH;
    }
    if (count != nullptr && liso_zero::zero_async(count, rows * sizeof(int), s) != hipSuccess) return LISO_ELAUNCH;
    if (c->n > 0) {
        const dim3 grid((unsigned)((c->n + kThreads - 1) / kThreads), (unsigned)((c->k + kTile - 1) / kTile), (unsigned)c->batch);
        const float* fl = mean_flow != nullptr ? flow : nullptr;
        if (c->precision == 0)
            hipLaunchKernelGGL(points_in_boxes_kernel<0>, grid, dim3(kThreads), 0, s, *c, boxes, points, point_valid, fl, mask,
                               count, fsum);
        else
            hipLaunchKernelGGL(points_in_boxes_kernel<1>, grid, dim3(kThreads), 0, s, *c, boxes, points, point_valid, fl, mask,
                               count, fsum);
    }
    if (mean_flow != nullptr)
        hipLaunchKernelGGL(mean_flow_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, (long)rows, count, fsum,
                           mean_flow);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

extern "C" int liso_match_greedy_f32(const float* iou, long gt_stride, long pred_stride, int n_gt, int n_pred,
                                     const int64_t* pred_order, float threshold,
                                     int64_t* idx_gt, int64_t* idx_pred, float* match_iou, int* num_matches,
                                     uint8_t* matched_pred_mask, uint8_t* detected_gt_mask, void* stream) {
    if (n_gt < 0 || n_pred < 0 || num_matches == nullptr) return LISO_EINVAL;
    if (n_pred > 0 && (pred_order == nullptr || matched_pred_mask == nullptr)) return LISO_EINVAL;
    if (n_gt > 0 && detected_gt_mask == nullptr) return LISO_EINVAL;
    if (n_gt > 0 && n_pred > 0 && (iou == nullptr || idx_gt == nullptr || idx_pred == nullptr || match_iou == nullptr))
        return LISO_EINVAL;
#define LISO_GREEDY(R)                                                                                                       \
    hipLaunchKernelGGL(greedy_match_kernel<R>, dim3(1), dim3(64), 0, (hipStream_t)stream, iou, gt_stride, pred_stride, n_gt, n_pred,                \
                       (const long long*)pred_order, threshold, (long long*)idx_gt, (long long*)idx_pred, match_iou, num_matches, \
                       matched_pred_mask, detected_gt_mask)
    if (n_gt <= 64) LISO_GREEDY(1);
    else if (n_gt <= 256) LISO_GREEDY(4);
    else if (n_gt <= 1024) LISO_GREEDY(16);
    else LISO_GREEDY(0);
#undef LISO_GREEDY
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}
