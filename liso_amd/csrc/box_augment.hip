// Box-snippet augmentation kernels for gfx950.  C ABI + reference lines: include/liso_augment.h.
//
// free mask: the sweep's occupancy is scattered into a byte map (plain stores), and each block packs the 2r + 1 rows around its
// BEV row into an LDS BITMAP.  A disk of radius r is, row by row, a horizontal run of half width floor(sqrt(r^2 - dy^2)); "any
// occupied cell in the run" is one or two masked 64-bit word tests, so a cell costs 2r + 1 LDS probes instead of the ~pi r^2
// footprint probes of a generic binary dilation (21 vs 317 at the 512^2 / 100 m set-up).  The row's count of free cells comes
// from one block reduction, and a single-block scan turns the counts into the prefix that `select` bisects.  HBM traffic = the n
// coordinate pairs once + h * w mask bytes once (the byte map and its 2r + 1-fold re-reads stay in L2: 256 KB at 512^2).
//
// paste: one block per pasted object (<= 15 per sample, a few hundred points each) -- the work is tiny, the point is that it
// happens where the sweep already lives, and in a fixed summation order (no float atomics).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_augment.h"
#include "../../include/liso_iou3d.h"
#include "zero_fill.h"
#include "per_device.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxRadius = 255;

// occupancy as a BYTE map with plain stores (every writer stores the same 1: no atomics, no ordering needed); rows are padded to
// a multiple of 64 cells so the mask kernel can read them as aligned 8-byte words
__global__ __launch_bounds__(kThreads) void occupancy_bytes_kernel(const int32_t* __restrict__ coors, long n, int h, int w, int wp,
                                                                   uint8_t* __restrict__ occ) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const int2 c = reinterpret_cast<const int2*>(coors)[i];
    if ((unsigned)c.x < (unsigned)h && (unsigned)c.y < (unsigned)w) occ[(long)c.x * wp + c.y] = 1;
}

// any bit of the row in columns [lo, hi] (0 <= lo <= hi < w)
__device__ __forceinline__ bool any_in_run(const unsigned long long* row, int lo, int hi) {
    const int wl = lo >> 6, wh = hi >> 6;
    const unsigned long long first = ~0ull << (lo & 63), last = ~0ull >> (63 - (hi & 63));
    if (wl == wh) return (row[wl] & first & last) != 0ull;
    if (row[wl] & first) return true;
    for (int j = wl + 1; j < wh; j++)
        if (row[j]) return true;
    return (row[wh] & last) != 0ull;
}

// One block per BEV row y: the 2r + 1 occupancy rows around y are packed into an LDS bitmap (8 cells per 8-byte load, one LDS
// atomicOr per octet), then every cell of the row probes its 2r + 1 runs in LDS.
__global__ __launch_bounds__(kThreads) void free_mask_kernel(const uint8_t* __restrict__ occ, int h, int w, int words, int radius,
                                                             uint8_t* __restrict__ free_mask, int* __restrict__ row_count) {
    extern __shared__ unsigned long long lds_bits[];  // [(2r + 1) rows][words]
    __shared__ int half_w[2 * kMaxRadius + 1];
    __shared__ int wave_sum[kThreads / 64];
    const int y = blockIdx.x;
    const int y_lo = y - radius < 0 ? 0 : y - radius, y_hi = y + radius >= h ? h - 1 : y + radius;
    const int rows = y_hi - y_lo + 1;
    for (int t = threadIdx.x; t < rows * words; t += kThreads) lds_bits[t] = 0ull;
    for (int t = threadIdx.x; t <= 2 * radius; t += kThreads) {
        const int dy = t - radius;
        int hw = (int)sqrtf((float)(radius * radius - dy * dy));  // largest dx with dx^2 + dy^2 <= r^2
        while ((hw + 1) * (hw + 1) + dy * dy <= radius * radius) hw++;
        while (hw * hw + dy * dy > radius * radius) hw--;
        half_w[t] = hw;
    }
    __syncthreads();
    const int octets = words * 8;  // 8-cell groups per row
    const unsigned long long* occ8 = reinterpret_cast<const unsigned long long*>(occ);
    for (int t = threadIdx.x; t < rows * octets; t += kThreads) {
        const int r = t / octets, o = t - r * octets;
        const unsigned long long v = occ8[(long)(y_lo + r) * octets + o];
        if (v) {
            // bytes are 0 / 1: gather bit 0 of each byte into 8 adjacent bits (byte k -> bit k)
            const unsigned long long packed = ((v & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
            atomicOr(&lds_bits[r * words + (o >> 3)], packed << ((o & 7) * 8));
        }
    }
    __syncthreads();
    int mine = 0;
    for (int x = threadIdx.x; x < w; x += kThreads) {
        bool occupied = false;
        for (int yy = y_lo; yy <= y_hi && !occupied; yy++) {
            const int hw = half_w[yy - y + radius];
            const int lo = x - hw < 0 ? 0 : x - hw, hi = x + hw >= w ? w - 1 : x + hw;
            occupied = any_in_run(lds_bits + (yy - y_lo) * words, lo, hi);
        }
        free_mask[(long)y * w + x] = occupied ? 0 : 1;
        mine += occupied ? 0 : 1;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int k = 0; k < kThreads / 64; k++) s += wave_sum[k];
        row_count[y] = s;
    }
}

// prefix[0] = 0, prefix[i + 1] = prefix[i] + count[i]; one block, chunks of kThreads rows with a running carry (in place:
// `count` aliases prefix + 1)
__global__ __launch_bounds__(kThreads) void row_prefix_kernel(int* __restrict__ prefix, int h) {
    __shared__ int buf[kThreads];
    __shared__ int carry;
    if (threadIdx.x == 0) {
        carry = 0;
        prefix[0] = 0;
    }
    __syncthreads();
    for (int base = 0; base < h; base += kThreads) {
        const int i = base + threadIdx.x;
        const int v = i < h ? prefix[i + 1] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < kThreads; o <<= 1) {
            const int add = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < h) prefix[i + 1] = carry + buf[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) carry += buf[kThreads - 1];
        __syncthreads();
    }
}

// one wavefront per query: bisect the row prefix, then each lane counts the free cells of its slice of the row, a wave scan finds
// the slice that holds the wanted cell and one lane walks that slice
__global__ __launch_bounds__(64) void select_free_cells_kernel(const uint8_t* __restrict__ free_mask, const int* __restrict__ prefix, int h,
                                                               int w, const int64_t* __restrict__ compact_idx, int k,
                                                               int* __restrict__ flat_cell) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int64_t want = compact_idx[q];
    if (want < 0 || want >= prefix[h]) {
        if (lane == 0) flat_cell[q] = -1;
        return;
    }
    int lo = 0, hi = h;  // largest row with prefix[row] <= want
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= want) lo = mid; else hi = mid;
    }
    const int left = (int)(want - prefix[lo]);
    const uint8_t* row = free_mask + (long)lo * w;
    const int per = (w + 63) / 64, x0 = lane * per, x1 = x0 + per < w ? x0 + per : w;
    int mine = 0;
    for (int x = x0; x < x1; x++) mine += row[x] ? 1 : 0;
    int incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const int before = incl - mine;
    if (left >= before && left < incl) {
        int need = left - before, x = x0;
        for (; x < x1; x++)
            if (row[x] && need-- == 0) break;
        flat_cell[q] = lo * w + x;
    }
}

__global__ __launch_bounds__(kThreads) void snippet_paste_kernel(const float4* __restrict__ db_points, long db_rows,
                                                                 const int64_t* __restrict__ src_index, const int64_t* __restrict__ out_offsets,
                                                                 const double* __restrict__ pose, const double* __restrict__ flow_rand,
                                                                 double vmin, double vmax, float4* __restrict__ out_points,
                                                                 float* __restrict__ out_flow, float* __restrict__ box_velo) {
    __shared__ double part[kThreads];
    const int obj = blockIdx.x;
    const long begin = out_offsets[obj], end = out_offsets[obj + 1];
    double m[12];
#pragma unroll
    for (int e = 0; e < 12; e++) m[e] = pose[obj * 12 + e];
    double speed = 0.0;
    for (long j = begin + threadIdx.x; j < end; j += kThreads) {
        const int64_t src = src_index[j];
        float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src >= 0 && src < db_rows) p = db_points[src];
        const double x = p.x, y = p.y, z = p.z;
        float4 o;
        // (the reference's einsum adds the four products left to right; no contraction, so the float64 values match)
        o.x = (float)(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m[0], x), __dmul_rn(m[1], y)), __dmul_rn(m[2], z)), m[3]));
        o.y = (float)(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m[4], x), __dmul_rn(m[5], y)), __dmul_rn(m[6], z)), m[7]));
        o.z = (float)(__dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(m[8], x), __dmul_rn(m[9], y)), __dmul_rn(m[10], z)), m[11]));
        o.w = p.w;
        out_points[j] = o;
        const double fx = __dadd_rn(vmin, __dmul_rn(flow_rand[3 * j + 0], vmax - vmin));
        const double fy = __dadd_rn(vmin, __dmul_rn(flow_rand[3 * j + 1], vmax - vmin));
        const double fz = __dadd_rn(vmin, __dmul_rn(flow_rand[3 * j + 2], vmax - vmin));
        if (out_flow) {
            out_flow[3 * j + 0] = (float)fx;
            out_flow[3 * j + 1] = (float)fy;
            out_flow[3 * j + 2] = (float)fz;
        }
        speed += sqrt(fx * fx + fy * fy + fz * fz);
    }
    part[threadIdx.x] = speed;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) box_velo[obj] = end > begin ? (float)(part[0] / (double)(end - begin)) : 0.0f;
}

int words_per_row(int w) { return (w + 63) / 64; }

}  // namespace

extern "C" {

size_t liso_bev_free_mask_workspace_bytes(int h, int w) {
    if (h <= 0 || w <= 0) return 0;
    return (size_t)h * words_per_row(w) * 64;  // the occupancy byte map, rows padded to 64 cells
}

int liso_bev_free_mask(const int32_t* pillar_coors, long n, int h, int w, int radius, uint8_t* free_mask, int32_t* row_free_prefix,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (h <= 0 || w <= 0 || n < 0 || radius < 0 || radius > kMaxRadius || !free_mask || !row_free_prefix || !workspace) return LISO_EINVAL;
    if (n > 0 && !pillar_coors) return LISO_EINVAL;
    if (((uintptr_t)pillar_coors & 7) || ((uintptr_t)workspace & 7)) return LISO_EINVAL;
    if (workspace_bytes < liso_bev_free_mask_workspace_bytes(h, w)) return LISO_EWORKSPACE;
    const int words = words_per_row(w);
    const size_t lds = (size_t)(2 * radius + 1) * words * sizeof(unsigned long long);
    if (lds > 120 * 1024) return LISO_EINVAL;  // (radius 255 on a 1024-wide grid = 65 KB)
    hipStream_t st = (hipStream_t)stream;
    static liso_dev::PerDeviceFlag attr_set;
    if (!liso_dev::lds_opt_in(attr_set, (const void*)free_mask_kernel, 120 * 1024)) return LISO_ELAUNCH;
    uint8_t* occ = (uint8_t*)workspace;
    if (liso_zero::zero_async(occ, liso_bev_free_mask_workspace_bytes(h, w), st) != hipSuccess) return LISO_ELAUNCH;
    if (n > 0) occupancy_bytes_kernel<<<(unsigned)((n + kThreads - 1) / kThreads), kThreads, 0, st>>>(pillar_coors, n, h, w, words * 64, occ);
    free_mask_kernel<<<h, kThreads, lds, st>>>(occ, h, w, words, radius, free_mask, row_free_prefix + 1);
    row_prefix_kernel<<<1, kThreads, 0, st>>>(row_free_prefix, h);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

int liso_bev_select_free_cells(const uint8_t* free_mask, const int32_t* row_free_prefix, int h, int w, const int64_t* compact_idx, int k,
                               int32_t* flat_cell, void* stream) {
    if (h <= 0 || w <= 0 || k < 0 || !free_mask || !row_free_prefix) return LISO_EINVAL;
    if (k == 0) return LISO_OK;
    if (!compact_idx || !flat_cell) return LISO_EINVAL;
    select_free_cells_kernel<<<k, 64, 0, (hipStream_t)stream>>>(free_mask, row_free_prefix, h, w, compact_idx, k, flat_cell);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

int liso_snippet_paste(const float* db_points, long db_rows, const int64_t* src_index, const int64_t* out_offsets, const double* pose,
                       const double* flow_rand, double vmin, double vmax, int k, float* out_points, float* out_flow, float* box_velo,
                       void* stream) {
    if (k < 0 || db_rows < 0) return LISO_EINVAL;
    if (k == 0) return LISO_OK;
    if (!db_points || !src_index || !out_offsets || !pose || !flow_rand || !out_points || !box_velo) return LISO_EINVAL;
    if (((uintptr_t)db_points | (uintptr_t)out_points) & 15) return LISO_EINVAL;
    snippet_paste_kernel<<<k, kThreads, 0, (hipStream_t)stream>>>((const float4*)db_points, db_rows, src_index, out_offsets, pose, flow_rand,
                                                                   vmin, vmax, (float4*)out_points, out_flow, box_velo);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // extern "C"
