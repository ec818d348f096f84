// Rotated-BEV IoU / overlap matrices and NMS for gfx950 (MI355X), C ABI in include/liso_iou3d.h.
//
// Semantics follow the reference's iou3d_nms (host twin iou3d_nms/src/iou3d_cpu.cpp:59-229,
// CUDA kernels iou3d_nms/src/iou3d_nms_kernel.cu:236-372, host greedy iou3d_nms/src/iou3d_nms.cpp:113-132);
// the structure is CDNA4-first and not a translation of those kernels:
//
//   * one wavefront == one 64-bit suppression word: a wave owns (row, 64 consecutive columns), each lane
//     tests one pair, __ballot() is the mask word.  The IoU/overlap matrices use the same mapping so the
//     64 results of a wave are one coalesced 256-B store.
//   * per-box work (sin/cos, rotated corners, in-box limits, a conservative bounding radius) is done once
//     per tile by one thread per box and parked in LDS, not once per pair.
//   * an exact early-out (bounding circles disjoint => the reference would find 0 polygon vertices and
//     return 0) removes the divergent polygon path for non-overlapping pairs.
//   * polygon vertices live in LDS laid out [slot][thread] (bank = thread, conflict-free for any slot), the
//     polar angle of each vertex is computed once, then the reference's bubble sort runs on the keys.
//   * the greedy sweep runs on the device: row-blocks of the bit-matrix are staged through LDS, one wave
//     resolves the 64-row diagonal dependency on scalar registers, all waves OR the kept rows into `remv`.
//     No D2H copy, no host loop, graph-capturable.
//
// Built with -ffp-contract=off: the geometry must round like the reference's host code (no FMA).
// sin/cos are evaluated in double and rounded to float, which matches glibc's sinf/cosf in practice.
#include <hip/hip_runtime.h>
#include "zero_fill.h"
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"

namespace {

constexpr int kCols = 64;        // columns per tile == wavefront width == bits per mask word
constexpr int kRows = 16;        // rows per tile
constexpr int kThreads = 256;    // 4 waves, each owns kRows/4 rows
constexpr int kMaxPoly = 16;     // reference: Point cross_points[16] (iou3d_cpu.cpp:168)
constexpr float kEps = 1e-8f;    // iou3d_cpu.cpp:38

// per-box derived quantities, SoA in LDS: g[field][box]
enum GeoField { G_CX, G_CY, G_LIMX, G_LIMY, G_COS, G_SIN, G_AREA, G_RAD, G_PX0, G_PX1, G_PX2, G_PX3, G_PY0, G_PY1, G_PY2, G_PY3, G_N };

struct Geo {
    float cx, cy, limx, limy, c, s, area, rad;
    float px[4], py[4];
};

struct PolyLds {
    float x[kMaxPoly][kThreads];
    float y[kMaxPoly][kThreads];
    float key[kMaxPoly][kThreads];
};

__device__ __forceinline__ Geo make_geo(const float* __restrict__ b) {
    Geo g;
    const float x = b[0], y = b[1], dx = b[3], dy = b[4], ang = b[6];
    // iou3d_cpu.cpp:134-140 half extents and the axis-aligned corners
    const float hx = dx / 2, hy = dy / 2;
    const float x1 = x - hx, y1 = y - hy, x2 = x + hx, y2 = y + hy;
    // iou3d_cpu.cpp:158-159; cos(-a)==cos(a), sin(-a)==-sin(a) covers :80 as well
    const double ad = (double)ang;
    g.c = (float)cos(ad);
    g.s = (float)sin(ad);
    g.cx = x;
    g.cy = y;
    // iou3d_cpu.cpp:85  box[3] / 2 + MARGIN
    g.limx = dx / 2 + 1e-2f;
    g.limy = dy / 2 + 1e-2f;
    g.area = dx * dy;  // iou3d_cpu.cpp:225
    const float rx[4] = {x1, x2, x2, x1};
    const float ry[4] = {y1, y1, y2, y2};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // iou3d_cpu.cpp:119-123 rotate_around_center
        g.px[k] = (rx[k] - x) * g.c + (ry[k] - y) * (-g.s) + x;
        g.py[k] = (rx[k] - x) * g.s + (ry[k] - y) * g.c + y;
    }
    // conservative radius: half diagonal + in-box margin + slack for fp32 rounding of far-away coordinates
    g.rad = sqrtf(hx * hx + hy * hy) * 1.001f + 0.02f + 1e-5f * (fabsf(x) + fabsf(y) + fabsf(hx) + fabsf(hy));
    return g;
}

__device__ __forceinline__ void store_geo(float (*g)[kCols], int i, const Geo& v) {
    g[G_CX][i] = v.cx; g[G_CY][i] = v.cy; g[G_LIMX][i] = v.limx; g[G_LIMY][i] = v.limy;
    g[G_COS][i] = v.c; g[G_SIN][i] = v.s; g[G_AREA][i] = v.area; g[G_RAD][i] = v.rad;
#pragma unroll
    for (int k = 0; k < 4; k++) { g[G_PX0 + k][i] = v.px[k]; g[G_PY0 + k][i] = v.py[k]; }
}

__device__ __forceinline__ Geo load_geo(const float (*g)[kCols], int i) {
    Geo v;
    v.cx = g[G_CX][i]; v.cy = g[G_CY][i]; v.limx = g[G_LIMX][i]; v.limy = g[G_LIMY][i];
    v.c = g[G_COS][i]; v.s = g[G_SIN][i]; v.area = g[G_AREA][i]; v.rad = g[G_RAD][i];
#pragma unroll
    for (int k = 0; k < 4; k++) { v.px[k] = g[G_PX0 + k][i]; v.py[k] = g[G_PY0 + k][i]; }
    return v;
}

// iou3d_cpu.cpp:63-65
__device__ __forceinline__ float cross3(float p1x, float p1y, float p2x, float p2y, float p0x, float p0y) {
    return (p1x - p0x) * (p2y - p0y) - (p2x - p0x) * (p1y - p0y);
}

// iou3d_cpu.cpp:30-36 (ternary min/max, not fminf/fmaxf)
__device__ __forceinline__ float rmin(float a, float b) { return a > b ? b : a; }
__device__ __forceinline__ float rmax(float a, float b) { return a > b ? a : b; }

// iou3d_cpu.cpp:88-117
__device__ __forceinline__ bool seg_isect(float p1x, float p1y, float p0x, float p0y, float q1x, float q1y, float q0x,
                                          float q0y, float& ax, float& ay) {
    // :67-73 check_rect_cross(p0, p1, q0, q1)
    const bool rc = rmin(p0x, p1x) <= rmax(q0x, q1x) && rmin(q0x, q1x) <= rmax(p0x, p1x) &&
                    rmin(p0y, p1y) <= rmax(q0y, q1y) && rmin(q0y, q1y) <= rmax(p0y, p1y);
    if (!rc) return false;
    const float s1 = cross3(q0x, q0y, p1x, p1y, p0x, p0y);
    const float s2 = cross3(p1x, p1y, q1x, q1y, p0x, p0y);
    const float s3 = cross3(p0x, p0y, q1x, q1y, q0x, q0y);
    const float s4 = cross3(q1x, q1y, p1x, p1y, q0x, q0y);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = cross3(q1x, q1y, p1x, p1y, p0x, p0y);
    if (fabsf(s5 - s1) > kEps) {
        ax = (s5 * q0x - s1 * q1x) / (s5 - s1);
        ay = (s5 * q0y - s1 * q1y) / (s5 - s1);
    } else {
        const float a0 = p0y - p1y, b0 = p1x - p0x, c0 = p0x * p1y - p1x * p0y;
        const float a1 = q0y - q1y, b1 = q1x - q0x, c1 = q0x * q1y - q1x * q0y;
        const float D = a0 * b1 - a1 * b0;
        ax = (b0 * c1 - b1 * c0) / D;
        ay = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

// iou3d_cpu.cpp:75-86 with cos(-h), sin(-h) folded: angle_cos = c, angle_sin = -s
__device__ __forceinline__ bool in_box(const Geo& box, float px, float py) {
    const float ac = box.c, as = -box.s;
    const float rx = (px - box.cx) * ac + (py - box.cy) * (-as);
    const float ry = (px - box.cx) * as + (py - box.cy) * ac;
    return fabsf(rx) < box.limx && fabsf(ry) < box.limy;
}

// iou3d_cpu.cpp:128-220.  A = row box ("box_a"), B = column box ("box_b").
__device__ float box_overlap_dev(const Geo& A, const Geo& B, PolyLds* __restrict__ P, int tid) {
    // exact early-out: disjoint bounding circles => no edge crossing and no corner inside the other box
    // (margin included in rad) => cnt == 0 => the reference returns fabs(0)/2.
    {
        const float ddx = A.cx - B.cx, ddy = A.cy - B.cy;
        const float rr = A.rad + B.rad;
        if (ddx * ddx + ddy * ddy > rr * rr) return 0.f;
    }
    int cnt = 0;
    float sx = 0.f, sy = 0.f;  // poly_center accumulator, :170-181
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int i1 = (i + 1) & 3;  // corners[4] = corners[0]
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int j1 = (j + 1) & 3;
            float hx, hy;
            if (seg_isect(A.px[i1], A.py[i1], A.px[i], A.py[i], B.px[j1], B.py[j1], B.px[j], B.py[j], hx, hy)) {
                if (cnt < kMaxPoly) { P->x[cnt][tid] = hx; P->y[cnt][tid] = hy; }
                sx = sx + hx;
                sy = sy + hy;
                cnt++;
            }
        }
    }
    // :184-195 corners of one box inside the other, interleaved b_k then a_k
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (in_box(A, B.px[k], B.py[k])) {
            sx = sx + B.px[k];
            sy = sy + B.py[k];
            if (cnt < kMaxPoly) { P->x[cnt][tid] = B.px[k]; P->y[cnt][tid] = B.py[k]; }
            cnt++;
        }
        if (in_box(B, A.px[k], A.py[k])) {
            sx = sx + A.px[k];
            sy = sy + A.py[k];
            if (cnt < kMaxPoly) { P->x[cnt][tid] = A.px[k]; P->y[cnt][tid] = A.py[k]; }
            cnt++;
        }
    }
    // cnt < 3: the shoelace fan below is empty or degenerate (cross with a zero vector) => 0
    if (cnt < 3) return 0.f;
    const float pcx = sx / cnt, pcy = sy / cnt;  // :197-198
    if (cnt > kMaxPoly) cnt = kMaxPoly;          // unreachable for convex quads (<= 8 crossings + 8 corners)

    // polar angle of every vertex once; point_cmp (:125-127) compares exactly these values
    for (int k = 0; k < cnt; k++) P->key[k][tid] = atan2f(P->y[k][tid] - pcy, P->x[k][tid] - pcx);

    // :199-209 bubble sort, swap when key[i] > key[i+1]
    for (int j = 0; j < cnt - 1; j++) {
        float kl = P->key[0][tid], xl = P->x[0][tid], yl = P->y[0][tid];
        for (int i = 0; i < cnt - j - 1; i++) {
            const float kr = P->key[i + 1][tid], xr = P->x[i + 1][tid], yr = P->y[i + 1][tid];
            if (kl > kr) {  // swap: right element moves to slot i, left one keeps bubbling
                P->key[i][tid] = kr; P->x[i][tid] = xr; P->y[i][tid] = yr;
            } else {
                P->key[i][tid] = kl; P->x[i][tid] = xl; P->y[i][tid] = yl;
                kl = kr; xl = xr; yl = yr;
            }
        }
        const int last = cnt - j - 1;
        P->key[last][tid] = kl; P->x[last][tid] = xl; P->y[last][tid] = yl;
    }

    // :211-217 shoelace fan about vertex 0
    const float x0 = P->x[0][tid], y0 = P->y[0][tid];
    float area = 0.f;
    float ux = P->x[0][tid] - x0, uy = P->y[0][tid] - y0;
    for (int k = 0; k < cnt - 1; k++) {
        const float vx = P->x[k + 1][tid] - x0, vy = P->y[k + 1][tid] - y0;
        area += ux * vy - uy * vx;
        ux = vx; uy = vy;
    }
    return fabsf(area) / 2.0f;
}

// iou3d_cpu.cpp:222-229
__device__ __forceinline__ float iou_from_overlap(const Geo& A, const Geo& B, float ov) {
    return ov / fmaxf(A.area + B.area - ov, kEps);
}

// iou3d_nms_kernel.cu:314-326
__device__ __forceinline__ float iou_normal_dev(const float* a, const float* b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, kEps);
}

struct TileLds {
    float col[G_N][kCols];
    float row[G_N][kCols];  // only kRows used; same shape keeps one load_geo
    PolyLds poly;
};

__device__ __forceinline__ void stage_tile(TileLds& L, const float* __restrict__ boxes_r, int nr, int r0,
                                           const float* __restrict__ boxes_c, int nc, int c0) {
    const int t = threadIdx.x;
    if (t < kCols) {
        const int c = c0 + t;
        if (c < nc) store_geo(L.col, t, make_geo(boxes_c + (size_t)c * 7));
    } else if (t < kCols + kRows) {
        const int r = r0 + (t - kCols);
        if (r < nr) store_geo(L.row, t - kCols, make_geo(boxes_r + (size_t)r * 7));
    }
    __syncthreads();
}

// MODE 0: overlap area (boxes_overlap_kernel, .cu:236-249); MODE 1: IoU (boxes_iou_bev_kernel, .cu:251-265)
template <int MODE>
__global__ __launch_bounds__(kThreads) void pair_matrix_kernel(const float* __restrict__ a, int n,
                                                                const float* __restrict__ b, int m,
                                                                float* __restrict__ out) {
    __shared__ TileLds L;
    const int c0 = blockIdx.x * kCols, r0 = blockIdx.y * kRows;
    stage_tile(L, a, n, r0, b, m, c0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = c0 + lane;
    if (c >= m) return;
    const Geo B = load_geo(L.col, lane);
#pragma unroll 1
    for (int rr = wave; rr < kRows; rr += kThreads / 64) {
        const int r = r0 + rr;
        if (r >= n) break;
        const Geo A = load_geo(L.row, rr);
        const float ov = box_overlap_dev(A, B, &L.poly, threadIdx.x);
        out[(size_t)r * m + c] = MODE == 0 ? ov : iou_from_overlap(A, B, ov);
    }
}

// suppression words, nms_kernel (.cu:267-311): bit i of mask[row][cb] <=> iou(row, cb*64+i) > thresh and col > row
__global__ __launch_bounds__(kThreads) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh,
                                                             unsigned long long* __restrict__ mask, int col_blocks) {
    const int c0 = blockIdx.x * kCols, r0 = blockIdx.y * kRows;
    if (c0 + kCols - 1 <= r0) return;  // tile entirely on/below the diagonal: never read by the greedy pass
    __shared__ TileLds L;
    stage_tile(L, boxes, n, r0, boxes, n, c0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = c0 + lane;
    Geo B;
    if (c < n) B = load_geo(L.col, lane);
#pragma unroll 1
    for (int rr = wave; rr < kRows; rr += kThreads / 64) {
        const int r = r0 + rr;
        if (r >= n) break;
        bool hit = false;
        if (c < n && c > r) {
            const Geo A = load_geo(L.row, rr);
            const float ov = box_overlap_dev(A, B, &L.poly, threadIdx.x);
            hit = iou_from_overlap(A, B, ov) > thresh;
        }
        const unsigned long long word = __ballot(hit);
        if (lane == 0) mask[(size_t)r * col_blocks + blockIdx.x] = word;
    }
}

// nms_normal_kernel (.cu:328-372)
__global__ __launch_bounds__(kThreads) void nms_normal_mask_kernel(const float* __restrict__ boxes, int n, float thresh,
                                                                    unsigned long long* __restrict__ mask,
                                                                    int col_blocks) {
    const int c0 = blockIdx.x * kCols, r0 = blockIdx.y * kRows;
    if (c0 + kCols - 1 <= r0) return;
    __shared__ float cb[kCols][7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < kCols * 7; i += kThreads) {
        const int g = c0 * 7 + i;
        cb[i / 7][i % 7] = g < n * 7 ? boxes[g] : 0.f;
    }
    __syncthreads();
    const int c = c0 + lane;
    for (int rr = wave; rr < kRows; rr += kThreads / 64) {
        const int r = r0 + rr;
        if (r >= n) break;
        bool hit = false;
        if (c < n && c > r) hit = iou_normal_dev(boxes + (size_t)r * 7, cb[lane]) > thresh;
        const unsigned long long word = __ballot(hit);
        if (lane == 0) mask[(size_t)r * col_blocks + blockIdx.x] = word;
    }
}

// Device twin of the host greedy sweep (iou3d_nms.cpp:113-132).  One workgroup.
//   remv[]   : LDS, one word per column block
//   per row-block b: stage mask rows [64b, 64b+64) x words [b, col_blocks) into LDS; wave 0 walks the 64 rows
//   serially on the diagonal word (the only true dependency), then every thread ORs the kept rows' words
//   into remv for the column blocks to the right.
constexpr int kGreedyThreads = 256;
constexpr int kGreedyMaxLdsWords = 6 * 1024;  // 48 KiB of tile: col_blocks <= 96 (n <= 6144) staged via LDS, else L2 reads

__global__ __launch_bounds__(kGreedyThreads) void nms_greedy_kernel(const unsigned long long* __restrict__ mask, int n,
                                                                    int col_blocks, long long* __restrict__ keep,
                                                                    int* __restrict__ num_out) {
    extern __shared__ unsigned long long lds[];
    unsigned long long* remv = lds;                 // [col_blocks]
    unsigned long long* tile = lds + col_blocks;    // [64][w] with w = col_blocks - b, or unused when !use_lds
    __shared__ unsigned long long kept_word;
    const bool use_lds = (size_t)col_blocks * 64 <= (size_t)kGreedyMaxLdsWords;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < col_blocks; i += kGreedyThreads) remv[i] = 0ULL;
    int total = 0;  // meaningful in wave 0 only
    __syncthreads();
    for (int b = 0; b < col_blocks; b++) {
        const int w = col_blocks - b;
        const int rbase = b * 64;
        if (use_lds) {
            for (int idx = tid; idx < 64 * w; idx += kGreedyThreads) {
                const int r = idx / w, cw = idx - r * w;
                const int row = rbase + r;
                tile[idx] = row < n ? mask[(size_t)row * col_blocks + b + cw] : 0ULL;
            }
            __syncthreads();
        }
        if (tid < 64) {
            const int row = rbase + lane;
            unsigned long long diag = 0ULL;
            if (row < n) diag = use_lds ? tile[lane * w] : mask[(size_t)row * col_blocks + b];
            const unsigned long long cur_v = remv[b];
            // wave-uniform: keep the serial chain on the scalar unit
            // (readfirstlane returns int: go through unsigned int so bit 31 is not sign-extended)
            const unsigned int cur_lo = (unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)cur_v);
            const unsigned int cur_hi = (unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(cur_v >> 32));
            unsigned long long cur = ((unsigned long long)cur_hi << 32) | (unsigned long long)cur_lo;
            const int valid = n - rbase;  // rows >= n are never kept
            if (valid < 64) cur |= ~0ULL << valid;
            unsigned long long kept = 0ULL;
#pragma unroll
            for (int t = 0; t < 64; t++) {
                const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((unsigned int)diag, t);
                const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((unsigned int)(diag >> 32), t);
                const unsigned long long d = ((unsigned long long)hi << 32) | lo;
                if (!((cur >> t) & 1ULL)) {
                    kept |= 1ULL << t;
                    cur |= d;
                }
            }
            if ((kept >> lane) & 1ULL) keep[total + __popcll(kept & ((1ULL << lane) - 1ULL))] = row;
            total += __popcll(kept);
            if (lane == 0) kept_word = kept;
        }
        __syncthreads();
        const unsigned long long kept = kept_word;
        for (int cw = 1 + tid; cw < w; cw += kGreedyThreads) {
            unsigned long long acc = remv[b + cw];
            unsigned long long k = kept;
            while (k) {
                const int t = __ffsll((long long)k) - 1;
                k &= k - 1;
                acc |= use_lds ? tile[t * w + cw] : mask[(size_t)(rbase + t) * col_blocks + b + cw];
            }
            remv[b + cw] = acc;
        }
        __syncthreads();
    }
    if (tid == 0) *num_out = total;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

template <int MODE>
int launch_pair_matrix(const float* a, int n, const float* b, int m, float* out, void* stream) {
    if (n < 0 || m < 0) return LISO_EINVAL;
    if (n == 0 || m == 0) return LISO_OK;
    if (!a || !b || !out) return LISO_EINVAL;
    dim3 grid((m + kCols - 1) / kCols, (n + kRows - 1) / kRows);
    hipLaunchKernelGGL(pair_matrix_kernel<MODE>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, n, b, m, out);
    return check_launch();
}

int launch_nms(bool normal, const float* boxes, int n, float thresh, int64_t* keep, int* num_out, void* ws,
               size_t ws_bytes, void* stream) {
    if (n < 0 || !num_out) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (liso_zero::zero_async(num_out, sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
        return LISO_OK;
    }
    if (!boxes || !keep || !ws) return LISO_EINVAL;
    if (ws_bytes < liso_iou3d_nms_workspace_bytes(n)) return LISO_EWORKSPACE;
    const int cb = (n + 63) / 64;
    auto* mask = (unsigned long long*)ws;
    dim3 grid(cb, (n + kRows - 1) / kRows);
    if (normal)
        hipLaunchKernelGGL(nms_normal_mask_kernel, grid, dim3(kThreads), 0, st, boxes, n, thresh, mask, cb);
    else
        hipLaunchKernelGGL(nms_mask_kernel, grid, dim3(kThreads), 0, st, boxes, n, thresh, mask, cb);
    if (check_launch() != LISO_OK) return LISO_ELAUNCH;
    const bool use_lds = (size_t)cb * 64 <= (size_t)kGreedyMaxLdsWords;
    const size_t lds_bytes = ((size_t)cb + (use_lds ? (size_t)cb * 64 : 0)) * sizeof(unsigned long long);
    hipLaunchKernelGGL(nms_greedy_kernel, dim3(1), dim3(kGreedyThreads), lds_bytes, st, mask, n, cb, (long long*)keep,
                       num_out);
    return check_launch();
}

// ---- host twin of boxes_iou_bev_cpu (iou3d_cpu.cpp:232-252): an explicitly-CPU entry point of the
// reference module, not a fallback for the device path ----
struct HPt { float x, y; };

inline float h_cross3(HPt p1, HPt p2, HPt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
inline float h_min(float a, float b) { return a > b ? b : a; }
inline float h_max(float a, float b) { return a > b ? a : b; }

inline bool h_isect(HPt p1, HPt p0, HPt q1, HPt q0, HPt& ans) {
    if (!(h_min(p0.x, p1.x) <= h_max(q0.x, q1.x) && h_min(q0.x, q1.x) <= h_max(p0.x, p1.x) &&
          h_min(p0.y, p1.y) <= h_max(q0.y, q1.y) && h_min(q0.y, q1.y) <= h_max(p0.y, p1.y)))
        return false;
    const float s1 = h_cross3(q0, p1, p0), s2 = h_cross3(p1, q1, p0), s3 = h_cross3(p0, q1, q0), s4 = h_cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    const float s5 = h_cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > kEps) {
        ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans.x = (b0 * c1 - b1 * c0) / D;
        ans.y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}

inline bool h_in_box(const float* box, HPt p) {
    const float c = cosf(-box[6]), s = sinf(-box[6]);
    const float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
    const float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
    return fabsf(rx) < box[3] / 2 + 1e-2f && fabsf(ry) < box[4] / 2 + 1e-2f;
}

inline void h_corners(const float* box, HPt out[5]) {
    const float hx = box[3] / 2, hy = box[4] / 2, c = cosf(box[6]), s = sinf(box[6]);
    const float xs[4] = {box[0] - hx, box[0] + hx, box[0] + hx, box[0] - hx};
    const float ys[4] = {box[1] - hy, box[1] - hy, box[1] + hy, box[1] + hy};
    for (int k = 0; k < 4; k++) {
        out[k].x = (xs[k] - box[0]) * c + (ys[k] - box[1]) * (-s) + box[0];
        out[k].y = (xs[k] - box[0]) * s + (ys[k] - box[1]) * c + box[1];
    }
    out[4] = out[0];
}

float h_iou_bev(const float* A, const float* B) {
    HPt ca[5], cb[5], poly[24];
    float key[24];
    h_corners(A, ca);
    h_corners(B, cb);
    int cnt = 0;
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            HPt h;
            if (h_isect(ca[i + 1], ca[i], cb[j + 1], cb[j], h)) { poly[cnt++] = h; sx = sx + h.x; sy = sy + h.y; }
        }
    for (int k = 0; k < 4; k++) {
        if (h_in_box(A, cb[k])) { sx = sx + cb[k].x; sy = sy + cb[k].y; poly[cnt++] = cb[k]; }
        if (h_in_box(B, ca[k])) { sx = sx + ca[k].x; sy = sy + ca[k].y; poly[cnt++] = ca[k]; }
    }
    float ov = 0.f;
    if (cnt >= 3) {
        const float pcx = sx / cnt, pcy = sy / cnt;
        for (int k = 0; k < cnt; k++) key[k] = atan2f(poly[k].y - pcy, poly[k].x - pcx);
        for (int j = 0; j < cnt - 1; j++)
            for (int i = 0; i < cnt - j - 1; i++)
                if (key[i] > key[i + 1]) {
                    const HPt t = poly[i]; poly[i] = poly[i + 1]; poly[i + 1] = t;
                    const float tk = key[i]; key[i] = key[i + 1]; key[i + 1] = tk;
                }
        float area = 0.f;
        for (int k = 0; k < cnt - 1; k++)
            area += (poly[k].x - poly[0].x) * (poly[k + 1].y - poly[0].y) - (poly[k].y - poly[0].y) * (poly[k + 1].x - poly[0].x);
        ov = fabsf(area) / 2.0f;
    }
    return ov / fmaxf(A[3] * A[4] + B[3] * B[4] - ov, kEps);
}

}  // namespace

extern "C" {

int liso_iou3d_iou_bev_cpu_f32(const float* a, int n, const float* b, int m, float* out) {
    if (n < 0 || m < 0) return LISO_EINVAL;
    if (n == 0 || m == 0) return LISO_OK;
    if (!a || !b || !out) return LISO_EINVAL;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++) out[(size_t)i * m + j] = h_iou_bev(a + (size_t)i * 7, b + (size_t)j * 7);
    return LISO_OK;
}

int liso_iou3d_overlap_bev_f32(const float* a, int n, const float* b, int m, float* out, void* stream) {
    return launch_pair_matrix<0>(a, n, b, m, out, stream);
}

int liso_iou3d_iou_bev_f32(const float* a, int n, const float* b, int m, float* out, void* stream) {
    return launch_pair_matrix<1>(a, n, b, m, out, stream);
}

size_t liso_iou3d_nms_workspace_bytes(int n) {
    if (n <= 0) return 0;
    return (size_t)n * (size_t)((n + 63) / 64) * sizeof(unsigned long long);
}

int liso_iou3d_nms_f32(const float* boxes, int n, float thresh, int64_t* keep_dev, int* num_out_dev, void* workspace,
                       size_t workspace_bytes, void* stream) {
    return launch_nms(false, boxes, n, thresh, keep_dev, num_out_dev, workspace, workspace_bytes, stream);
}

int liso_iou3d_nms_normal_f32(const float* boxes, int n, float thresh, int64_t* keep_dev, int* num_out_dev,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return launch_nms(true, boxes, n, thresh, keep_dev, num_out_dev, workspace, workspace_bytes, stream);
}

}  // extern "C"
