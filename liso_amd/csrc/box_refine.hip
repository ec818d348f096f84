// Local box refinement of the tracker for gfx950: the "closeness to edge" rectangle fit of the sweep points inside the bloated BEV
// footprint of each box.  C ABI + reference lines: include/liso_tracking.h (liso_fit_boxes_closeness_f32).
//
// One block per box, three phases:
//   1. ordered compaction: the block walks the sweep in chunks of 256 points; points whose box-frame x, y (fp64 transform, compared
//      as fp32 like the reference) lie inside 0.5 * bloat * dims are appended to the box's list in sweep order (ballot + popcount
//      prefix: the list -- and with it every floating-point sum below -- does not depend on scheduling);
//   2. for each of the 19 candidate headings (0, 5, ..., 90 deg): bounding extents of the projected list (block min / max), then
//      the closeness score sum 1 / max(distance to the nearest edge, d0) in a fixed order; first maximum wins;
//   3. extents at the winning heading (turned by 90 deg when the y side is the longer one), centre / length / width from the corners.
// The work is small (a box holds a few hundred points of a 120k-point sweep); the point is that it runs where the sweep lives and for
// all boxes of a frame at once, instead of one numpy loop per box and time step.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_tracking.h"

namespace {

constexpr int kThreads = 256;
constexpr int kAngles = 19;
constexpr double kD0 = 1e-2;
constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

struct Extents {
    double x0, x1, y0, y1;
};

// extents of the projected list; every thread returns the block-wide result
__device__ Extents block_extents(const double2* __restrict__ list, int n, double c, double s, double (*red)[4]) {
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const double2 p = list[i];
        const double px = p.x * c + p.y * s, py = -p.x * s + p.y * c;
        x0 = fmin(x0, px); x1 = fmax(x1, px); y0 = fmin(y0, py); y1 = fmax(y1, py);
    }
    x0 = wave_min(x0); x1 = wave_max(x1); y0 = wave_min(y0); y1 = wave_max(y1);
    __syncthreads();  // (red is reused between calls)
    if ((threadIdx.x & 63) == 0) {
        double* r = red[threadIdx.x >> 6];
        r[0] = x0; r[1] = x1; r[2] = y0; r[3] = y1;
    }
    __syncthreads();
    Extents e = {red[0][0], red[0][1], red[0][2], red[0][3]};
    for (int w = 1; w < kThreads / 64; w++) {
        e.x0 = fmin(e.x0, red[w][0]); e.x1 = fmax(e.x1, red[w][1]); e.y0 = fmin(e.y0, red[w][2]); e.y1 = fmax(e.y1, red[w][3]);
    }
    return e;
}

// sum over the list in a fixed order: thread-strided partial sums, then a tree over the 256 partials
__device__ double block_closeness(const double2* __restrict__ list, int n, double c, double s, const Extents& e, double* part) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) {
        const double2 p = list[i];
        const double px = p.x * c + p.y * s, py = -p.x * s + p.y * c;
        const double dx = fmin(px - e.x0, e.x1 - px), dy = fmin(py - e.y0, e.y1 - py);
        acc += 1.0 / fmax(fmin(dx, dy), kD0);
    }
    __syncthreads();
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    return part[0];
}

__global__ __launch_bounds__(kThreads) void fit_boxes_closeness_kernel(const float* __restrict__ points, long n, int stride,
                                                                       const uint8_t* __restrict__ point_valid,
                                                                       const float* __restrict__ boxes, float half_bloat,
                                                                       double2* __restrict__ lists, int* __restrict__ count,
                                                                       double* __restrict__ fit) {
    __shared__ double red[kThreads / 64][4];
    __shared__ double part[kThreads];
    __shared__ int wave_cnt[kThreads / 64];
    __shared__ int total;
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* b = boxes + (long)k * 7;
    const double bx = b[0], by = b[1];
    const double yaw = (double)b[6];
    const double cy = cos(yaw), sy = sin(yaw);
    const float hx = b[3] * half_bloat, hy = b[4] * half_bloat;  // fp32, like dims * (0.5 * bloat) on a float32 tensor
    double2* list = lists + (long)k * n;
    if (tid == 0) total = 0;
    __syncthreads();
    // ---- 1. ordered compaction ---------------------------------------------------------------------------------------------
    for (long base = 0; base < n; base += kThreads) {
        const long i = base + tid;
        bool in = false;
        double px = 0.0, py = 0.0;
        if (i < n && (!point_valid || point_valid[i])) {
            px = (double)points[i * stride];
            py = (double)points[i * stride + 1];
            const double dx = px - bx, dy = py - by;
            const float fx = (float)(cy * dx + sy * dy), fy = (float)(-sy * dx + cy * dy);
            in = fabsf(fx) < hx && fabsf(fy) < hy;  // (NaN coordinates compare false)
        }
        const unsigned long long m = __ballot(in);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wave; w++) off += wave_cnt[w];
        if (in) list[off + __popcll(m & ((1ull << lane) - 1ull))] = make_double2(px, py);
        __syncthreads();
        if (tid == 0) total += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int cnt = total;
    if (tid == 0) count[k] = cnt;
    if (cnt == 0) {
        if (tid < 5) fit[(long)k * 5 + tid] = nan("");
        return;
    }
    __threadfence_block();
    // ---- 2. the heading with the largest closeness score ---------------------------------------------------------------------
    double best = -INFINITY, angle = 0.0;
    for (int a = 0; a < kAngles; a++) {
        const double ang = (double)(a * 5) / 180.0 * kPi;
        const double c = cos(ang), s = sin(ang);
        const Extents e = block_extents(list, cnt, c, s, red);
        const double beta = block_closeness(list, cnt, c, s, e, part);
        if (beta > best) {
            best = beta;
            angle = ang;
        }
        __syncthreads();
    }
    // ---- 3. the rectangle at that heading ----------------------------------------------------------------------------------------
    double c = cos(angle), s = sin(angle);
    Extents e = block_extents(list, cnt, c, s, red);
    if ((e.x1 - e.x0) < (e.y1 - e.y0)) {
        angle = angle + kPi / 2;
        c = cos(angle);
        s = sin(angle);
        e = block_extents(list, cnt, c, s, red);
    }
    if (tid == 0) {
        // corners (x1,y0), (x0,y0), (x0,y1), (x1,y1) in the rotated frame times [[c, s], [-s, c]]
        const double c0x = e.x1 * c - e.y0 * s, c0y = e.x1 * s + e.y0 * c;
        const double c1x = e.x0 * c - e.y0 * s, c1y = e.x0 * s + e.y0 * c;
        const double c2x = e.x0 * c - e.y1 * s, c2y = e.x0 * s + e.y1 * c;
        const double c3x = e.x1 * c - e.y1 * s, c3y = e.x1 * s + e.y1 * c;
        double* o = fit + (long)k * 5;
        o[0] = (c0x + c2x) / 2;
        o[1] = (c0y + c2y) / 2;
        o[2] = sqrt((c0x - c1x) * (c0x - c1x) + (c0y - c1y) * (c0y - c1y));
        o[3] = sqrt((c0x - c3x) * (c0x - c3x) + (c0y - c3y) * (c0y - c3y));
        o[4] = angle;
    }
}

}  // namespace

extern "C" {

size_t liso_fit_boxes_closeness_workspace_bytes(long n, int k) {
    if (n <= 0 || k <= 0) return 0;
    return (size_t)n * (size_t)k * sizeof(double2);
}

int liso_fit_boxes_closeness_f32(const float* points, long n, int point_stride, const uint8_t* point_valid, const float* boxes, int k,
                                 float dims_bloat, int* count, double* fit, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || k < 0 || point_stride < 2) return LISO_EINVAL;
    if (k == 0) return LISO_OK;
    if (!boxes || !count || !fit || (n > 0 && (!points || !workspace))) return LISO_EINVAL;
    if (workspace_bytes < liso_fit_boxes_closeness_workspace_bytes(n, k) || ((uintptr_t)workspace & 15)) return LISO_EWORKSPACE;
    fit_boxes_closeness_kernel<<<k, kThreads, 0, (hipStream_t)stream>>>(points, n, point_stride, point_valid, boxes,
                                                                        (float)(0.5 * (double)dims_bloat), (double2*)workspace, count, fit);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // extern "C"
