// Exact 1-nearest-neighbour search on a uniform xy grid for gfx950.  C ABI + reference lines: include/liso_slim.h.
//
//   knn_count    one thread per reference point: cell id, atomicAdd(count[cell])
//   knn_scan     one 1024-thread block: exclusive scan of the cell counts (<= 1M cells)
//   knn_fill     one thread per reference point: position = start[cell] + atomicAdd(cursor[cell]); writes the point
//                (x, y, z, original index) into the bucketed array -> queries read 16 contiguous bytes per candidate
//   knn_query    one thread per query: ring 0, 1, 2, ... of cells around the query's cell; after ring r every
//                unvisited point is at least r*cell + (distance to the own cell's nearest edge) away, so the search
//                stops as soon as best <= that bound (exact), or when the rings cover the whole grid.
// LiDAR clouds are thin in z, so a 2-D grid with the full 3-D distance test is both exact and compact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ int cell_of(const liso_knn_grid& g, float x, float y, int* cx, int* cy) {
    *cx = clampi((int)floorf((x - g.x_min) / g.cell), 0, g.nx - 1);
    *cy = clampi((int)floorf((y - g.y_min) / g.cell), 0, g.ny - 1);
    return *cx * g.ny + *cy;
}

__global__ void knn_count_kernel(liso_knn_grid g, const float* __restrict__ ref, int stride, int n, int* __restrict__ count,
                                 int* __restrict__ cell_of_pt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cx, cy;
    const int c = cell_of(g, ref[(size_t)i * stride], ref[(size_t)i * stride + 1], &cx, &cy);
    cell_of_pt[i] = c;
    atomicAdd(&count[c], 1);
}

__global__ __launch_bounds__(1024) void knn_scan_kernel(const int* __restrict__ count, int cells, int* __restrict__ start) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (cells + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < cells ? lo + per : cells;
    int s = 0;
    for (int i = lo; i < hi; i++) s += count[i];
    part[tid] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 partials
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = tid == 0 ? 0 : part[tid - 1];
    for (int i = lo; i < hi; i++) { start[i] = run; run += count[i]; }
    if (tid == 1023) start[cells] = part[1023];
}

__global__ void knn_fill_kernel(const float* __restrict__ ref, int stride, int n, const int* __restrict__ cell_of_pt,
                                const int* __restrict__ start, int* __restrict__ cursor, float4* __restrict__ bucketed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cell_of_pt[i];
    const int pos = start[c] + atomicAdd(&cursor[c], 1);
    bucketed[pos] = make_float4(ref[(size_t)i * stride], ref[(size_t)i * stride + 1], ref[(size_t)i * stride + 2],
                                __int_as_float(i));
}

__global__ void knn_query_kernel(liso_knn_grid g, const int* __restrict__ start, const float4* __restrict__ bucketed,
                                 int n_ref, const float* __restrict__ query, int qstride, int nq,
                                 long long* __restrict__ index, float* __restrict__ dist_sqr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float qx = query[(size_t)i * qstride], qy = query[(size_t)i * qstride + 1], qz = query[(size_t)i * qstride + 2];
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz)) || n_ref == 0) {
        index[i] = 0;
        if (dist_sqr) dist_sqr[i] = nanf("");
        return;
    }
    int cx, cy;
    cell_of(g, qx, qy, &cx, &cy);
    // distance from the query to the edges of its (clamped) cell; a query outside the grid has a negative margin on that
    // side, which only makes the stopping bound more conservative (still exact)
    const float ox = qx - (g.x_min + cx * g.cell), oy = qy - (g.y_min + cy * g.cell);
    const float margin = fminf(fminf(ox, g.cell - ox), fminf(oy, g.cell - oy));
    float best = INFINITY;
    int best_i = 0x7fffffff;
    const int rmax = max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy));
    for (int r = 0; r <= rmax; r++) {
        const int x0 = cx - r, x1 = cx + r, y0 = cy - r, y1 = cy + r;
        for (int x = max(x0, 0); x <= min(x1, g.nx - 1); x++) {
            const bool edge_col = (x == x0 || x == x1);
            // on the ring: the full column for the two edge columns, else only the two end cells
            const int step = edge_col ? 1 : (y1 - y0 > 0 ? y1 - y0 : 1);
            for (int y = y0; y <= y1; y += step) {
                if (y < 0 || y >= g.ny) continue;
                const int c = x * g.ny + y;
                const int s = start[c], e = start[c + 1];
                for (int k = s; k < e; k++) {
                    const float4 p = bucketed[k];
                    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                    const float d = dx * dx + dy * dy + dz * dz;
                    const int pi = __float_as_int(p.w);
                    if (d < best || (d == best && pi < best_i)) { best = d; best_i = pi; }
                }
            }
        }
        const float bound = r * g.cell + margin;  // every unvisited point is at least this far (xy distance)
        if (bound > 0.f && best <= bound * bound) break;
    }
    index[i] = best_i == 0x7fffffff ? 0 : best_i;
    if (dist_sqr) dist_sqr[i] = best;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }
inline bool grid_ok(const liso_knn_grid* g) {
    return g && g->cell > 0.f && g->nx >= 1 && g->ny >= 1 && (long)g->nx * g->ny <= (1L << 20);
}
// workspace: count[cells] | cursor[cells] | start[cells+1] | cell_of_pt[n] | (pad to 16 B) bucketed[n] float4
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

extern "C" {

size_t liso_knn_workspace_bytes(const liso_knn_grid* grid, int n_ref) {
    if (!grid_ok(grid) || n_ref < 0) return 0;
    const size_t cells = (size_t)grid->nx * grid->ny;
    return align16((3 * cells + 1 + (size_t)n_ref) * sizeof(int)) + (size_t)n_ref * sizeof(float4) + 16;
}

int liso_knn_build_f32(const liso_knn_grid* grid, const float* ref, int ref_stride, int n_ref, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!grid_ok(grid) || n_ref < 0 || ref_stride < 3 || !workspace || (n_ref > 0 && !ref)) return LISO_EINVAL;
    if (workspace_bytes < liso_knn_workspace_bytes(grid, n_ref)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int cells = grid->nx * grid->ny;
    int* count = (int*)workspace;
    int* cursor = count + cells;
    int* start = cursor + cells;
    int* cell_of_pt = start + cells + 1;
    float4* bucketed = (float4*)((char*)workspace + align16((3 * (size_t)cells + 1 + (size_t)n_ref) * sizeof(int)));
    if (hipMemsetAsync(count, 0, 2 * (size_t)cells * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (n_ref > 0) knn_count_kernel<<<(n_ref + 255) / 256, 256, 0, st>>>(*grid, ref, ref_stride, n_ref, count, cell_of_pt);
    knn_scan_kernel<<<1, 1024, 0, st>>>(count, cells, start);
    if (n_ref > 0) knn_fill_kernel<<<(n_ref + 255) / 256, 256, 0, st>>>(ref, ref_stride, n_ref, cell_of_pt, start, cursor, bucketed);
    return check_launch();
}

int liso_knn_query_f32(const liso_knn_grid* grid, const float* ref, int ref_stride, int n_ref, const void* workspace,
                       const float* query, int query_stride, int n_query, int64_t* index, float* dist_sqr, void* stream) {
    (void)ref; (void)ref_stride;
    if (!grid_ok(grid) || n_ref < 0 || n_query < 0 || query_stride < 3 || !workspace) return LISO_EINVAL;
    if (n_query == 0) return LISO_OK;
    if (!query || !index) return LISO_EINVAL;
    const int cells = grid->nx * grid->ny;
    const int* start = (const int*)workspace + 2 * (size_t)cells;
    const float4* bucketed = (const float4*)((const char*)workspace + align16((3 * (size_t)cells + 1 + (size_t)n_ref) * sizeof(int)));
    knn_query_kernel<<<(n_query + 127) / 128, 128, 0, (hipStream_t)stream>>>(*grid, start, bucketed, n_ref, query, query_stride,
                                                                            n_query, (long long*)index, dist_sqr);
    return check_launch();
}

}  // extern "C"
