// Exact 1-nearest-neighbour search on a uniform grid for gfx950.  C ABI + reference lines: include/liso_slim.h.
//
// Bucket key = (xy cell, z bin): the reference points are counting-sorted by key, so the points of one xy cell are
// contiguous AND ordered by z bin.  LiDAR clouds are thin in z but full of vertical structure (a wall stacks hundreds of
// returns into one pillar), so the search walks rings of xy cells and, inside every cell, only reads the z bins that
// can still hold a closer point than the best found so far.
//
//   knn_count    one thread per reference point: key, atomicAdd(count[key])
//   knn_scan_*   exclusive scan of the key counts (per-1024 block scan, scan of the block totals, add)
//   knn_fill     one thread per reference point: position = start[key] + atomicAdd(cursor[key]); writes the point
//                (x, y, z, original index) into the bucketed array -> queries read 16 contiguous bytes per candidate
//   knn_query    16 lanes per query: ring 0, 1, 2, ... of cells around the query's cell; after ring r every
//                unvisited point is at least r*cell + (distance to the own cell's nearest edge) away, so the search
//                stops as soon as best <= that bound (exact), or when the rings cover the whole grid.
#include <hip/hip_runtime.h>
#include "zero_fill.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ int cell_of(const liso_knn_grid& g, float x, float y, int* cx, int* cy) {
    *cx = clampi((int)floorf((x - g.x_min) / g.cell), 0, g.nx - 1);
    *cy = clampi((int)floorf((y - g.y_min) / g.cell), 0, g.ny - 1);
    return *cx * g.ny + *cy;
}

// z bin, clamped: points below / above the binned range live in the first / last bin
__device__ __forceinline__ int zbin_of(const liso_knn_grid& g, float z) {
    return clampi((int)floorf((z - g.z_min) / g.z_cell), 0, g.nz - 1);
}

// Lanes of a wave that fall into the same bucket are served by ONE atomic: consecutive LiDAR returns hit the same 0.2 m cell, and
// 64 single atomics on one counter serialise (68 + 70 us for 120k points before; the loop below runs once per distinct bucket of the
// wave).  `rank` = position of this lane among the wave's lanes with the same key, `n` = how many there are, `leader` = the first.
__device__ __forceinline__ void wave_group_by_key(int key, bool active, int lane, int& rank, int& n, bool& leader) {
    rank = 0;
    n = 0;
    leader = false;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, src);
        const unsigned long long same = __ballot(active && key == k);
        if (active && key == k) {
            rank = __popcll(same & ((1ull << lane) - 1ull));
            n = __popcll(same);
            leader = lane == src;
        }
        todo &= ~same;
    }
}

__global__ __launch_bounds__(256) void knn_count_kernel(liso_knn_grid g, const float* __restrict__ ref, int stride, int n, int* __restrict__ count,
                                                        int* __restrict__ cell_of_pt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int c = -1;
    if (i < n) {
        int cx, cy;
        const float x = ref[(size_t)i * stride], y = ref[(size_t)i * stride + 1], z = ref[(size_t)i * stride + 2];
        if (isfinite(x) && isfinite(y) && isfinite(z)) c = cell_of(g, x, y, &cx, &cy) * g.nz + zbin_of(g, z);  // NaN padding rows: never neighbours
        cell_of_pt[i] = c;
    }
    int rank, cnt;
    bool leader;
    wave_group_by_key(c, c >= 0, threadIdx.x & 63, rank, cnt, leader);
    if (leader) atomicAdd(&count[c], cnt);
}

// exclusive scan of the cell counts in three short launches: per-1024-cell block scan, scan of the block totals, add
__global__ __launch_bounds__(1024) void knn_scan_block_kernel(const int* __restrict__ count, int cells, int* __restrict__ start,
                                                              int* __restrict__ block_tot) {
    __shared__ int wsum[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = i < cells ? count[i] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (i < cells) start[i] = base + incl - v;
    if (threadIdx.x == 1023) block_tot[blockIdx.x] = base + incl;
}

__global__ __launch_bounds__(1024) void knn_scan_tot_kernel(int* __restrict__ block_tot, int nblocks) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nblocks; c0 += 1024) {  // chunks of 1024 block totals, running carry
        const int i = c0 + (int)threadIdx.x;
        const int v = i < nblocks ? block_tot[i] : 0;
        int incl = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; w++) base += wsum[w];
        if (i < nblocks) block_tot[i] = base + incl - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_tot[nblocks] = carry;  // grand total
}

__global__ __launch_bounds__(1024) void knn_scan_add_kernel(int* __restrict__ start, int cells, const int* __restrict__ block_tot,
                                                            int nblocks) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < cells) start[i] += block_tot[blockIdx.x];
    if (i == 0) start[cells] = block_tot[nblocks];
}

__global__ __launch_bounds__(256) void knn_fill_kernel(const float* __restrict__ ref, int stride, int n, const int* __restrict__ cell_of_pt,
                                                       const int* __restrict__ start, int* __restrict__ cursor, float4* __restrict__ bucketed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int c = i < n ? cell_of_pt[i] : -1;
    // one slot reservation per (wave, bucket): the first lane of every group of equal keys adds the group's size to the bucket's
    // cursor and hands the base to the others
    int slot = 0;
    unsigned long long todo = __ballot(c >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const int k = __shfl(c, src);
        const unsigned long long same = __ballot(c == k);
        int base = 0;
        if (lane == src) base = atomicAdd(&cursor[k], __popcll(same));
        base = __shfl(base, src);
        if (c == k) slot = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (c < 0) return;
    bucketed[start[c] + slot] = make_float4(ref[(size_t)i * stride], ref[(size_t)i * stride + 1], ref[(size_t)i * stride + 2],
                                            __int_as_float(i));
}

__global__ void knn_sorted_ids_kernel(const float4* __restrict__ bucketed, int n, long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __float_as_int(bucketed[i].w);
}

// 16 lanes cooperate on one query: each lane fetches the [start, end) range of one cell of the ring (independent loads
// in flight), then the group walks every non-empty range together, 16 consecutive float4 per step (256 coalesced
// bytes) -- vertical structures put hundreds of points into one xy cell, so points, not cells, are the unit of work.
// A 4-step butterfly picks the group's best (distance, index) once per ring.
constexpr int kGroup = 16;  // lanes per query (measured: 8 lanes per query -> 1.24 ms per step instead of 1.02: more divergence, more batches)

struct Level {
    liso_knn_grid g;
    const int* start;
    const float4* bucketed;
    const unsigned* occ;  // one bit per xy cell: does the cell hold any point (any z bin)?
};

// after the scan the `count` array is dead: its first nx*ny/32 words become the occupancy bitmap (31 KB for a 500x500
// grid: cache resident), so that a ring walk reads the 8-byte [start, end) pair only for the few cells that hold points
__global__ void knn_occupancy_kernel(liso_knn_grid g, const int* __restrict__ start, unsigned* __restrict__ occ) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;  // xy cell; blockDim is a multiple of 64
    const int n = g.nx * g.ny;
    const bool full = c < n && start[(size_t)(c + 1) * g.nz] > start[(size_t)c * g.nz];
    const unsigned long long m = __ballot(full);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && c < n) occ[c >> 5] = (unsigned)m;
    if (lane == 32 && c < n) occ[c >> 5] = (unsigned)(m >> 32);
}

// ring search on one grid level; (best, best_i) carry over between levels (a candidate found on the fine level is a real
// point, so it only tightens the coarse search).  Returns true when the answer is proven exact.
__device__ __forceinline__ bool ring_search(const Level& L, float qx, float qy, float qz, int max_rings, int sub, int group_shift,
                                            float& best, int& best_i) {
    const liso_knn_grid& g = L.g;
    int cx, cy;
    cell_of(g, qx, qy, &cx, &cy);
    const float ox = qx - (g.x_min + cx * g.cell), oy = qy - (g.y_min + cy * g.cell);
    const float margin = fminf(fminf(ox, g.cell - ox), fminf(oy, g.cell - oy));
    const int rfull = max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy));
    const int rmax = (max_rings >= 0 && rfull > max_rings) ? max_rings : rfull;
    for (int r = 0; r <= rmax; r++) {
        const int ncell = r == 0 ? 1 : 8 * r;
        const int x0 = cx - r, x1 = cx + r, y0 = cy - r, y1 = cy + r;
        // only z bins that can hold a point closer than the best so far (|dz| <= sqrt(best)); everything before a first hit
        const float dzmax = sqrtf(best) * 1.0001f + 1e-6f;  // padded against sqrt rounding
        const int zlo = best == INFINITY ? 0 : zbin_of(g, qz - dzmax);
        const int zhi = best == INFINITY ? g.nz - 1 : zbin_of(g, qz + dzmax);
        for (int t0 = 0; t0 < ncell; t0 += kGroup) {
            const int t = t0 + sub;
            int s = 0, e = 0;
            if (t < ncell) {
                int x = cx, y = cy;
                if (r > 0) {
                    const int two_r = 2 * r;  // side = t / (2r) in {0,1,2,3} without an integer division
                    const int side = (t >= two_r) + (t >= 2 * two_r) + (t >= 3 * two_r), k = t - side * two_r;
                    x = side == 0 ? x0 + k : (side == 1 ? x1 : (side == 2 ? x1 - k : x0));
                    y = side == 0 ? y0 : (side == 1 ? y0 + k : (side == 2 ? y1 : y1 - k));
                }
                if (x >= 0 && x < g.nx && y >= 0 && y < g.ny) {
                    const int cxy = x * g.ny + y;
                    // no point of this cell can be closer (in xy) than the cell's nearest edge: skip it when that already exceeds the
                    // best so far (strictly: an equal distance may still win the tie on the index)
                    // (border cells also hold the points beyond the grid, cell_of clamps: their outer edge is at infinity)
                    const float ex = fmaxf(fmaxf(x == 0 ? 0.f : g.x_min + x * g.cell - qx, x == g.nx - 1 ? 0.f : qx - (g.x_min + (x + 1) * g.cell)), 0.f);
                    const float ey = fmaxf(fmaxf(y == 0 ? 0.f : g.y_min + y * g.cell - qy, y == g.ny - 1 ? 0.f : qy - (g.y_min + (y + 1) * g.cell)), 0.f);
                    if (!(ex * ex + ey * ey > best) && ((L.occ[cxy >> 5] >> (cxy & 31)) & 1u)) {
                        s = L.start[cxy * g.nz + zlo];
                        e = L.start[cxy * g.nz + zhi + 1];
                    }
                }
            }
            unsigned nonempty = (unsigned)(__ballot(e > s) >> group_shift) & ((1u << kGroup) - 1u);
            while (nonempty) {
                const int j = __ffs(nonempty) - 1;
                nonempty &= nonempty - 1;
                const int sj = __shfl(s, j, kGroup), ej = __shfl(e, j, kGroup);
                for (int k = sj + sub; k < ej; k += kGroup) {
                    const float4 p = L.bucketed[k];
                    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
                    const float d = dx * dx + dy * dy + dz * dz;
                    const int pi = __float_as_int(p.w);
                    if (d < best || (d == best && pi < best_i)) { best = d; best_i = pi; }
                }
            }
        }
        // group-wide best (ties -> smaller index).  The 16-lane group is one DPP row: quad_perm [1,0,3,2], quad_perm
        // [2,3,0,1], row_ror:4, row_ror:8 leave every lane with the row minimum -- VALU-rate moves instead of 8 ds_bpermute
        // round trips through the LDS crossbar per ring (min over (distance, index) is order independent).
        static_assert(kGroup == 8 || kGroup == 16, "the DPP reduction below assumes 8- or 16-lane groups inside one DPP row");
#define LISO_KNN_DPP_STEP(CTRL)                                                                                     \
        {                                                                                                           \
            const float ob = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(best), CTRL, 0xf, 0xf, false)); \
            const int oi = __builtin_amdgcn_mov_dpp(best_i, CTRL, 0xf, 0xf, false);                                 \
            if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }                              \
        }
        LISO_KNN_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]
        LISO_KNN_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]: every quad now holds its minimum in all four lanes
        if (kGroup == 8) {
            LISO_KNN_DPP_STEP(0x141)  // row_half_mirror: the other quad of the same 8 lanes
        } else {
            LISO_KNN_DPP_STEP(0x124)  // row_ror:4
            LISO_KNN_DPP_STEP(0x128)  // row_ror:8
        }
#undef LISO_KNN_DPP_STEP
        const float bound = r * g.cell + margin;  // every unvisited point is at least this far (xy distance)
        if (bound > 0.f && best <= bound * bound) return true;
    }
    return rmax == rfull;  // grid exhausted
}

// One launch answers every query: the fine level resolves the dense near field in a few rings; a group whose answer is
// not proven after `fine_max_rings` continues on the coarse level (no second launch, no pass over resolved rows).
// Counters (720k queries in bucket order, rocprofv3 --pmc, round 3): 1015 VALU + 660 SALU instructions and 37 vector loads per wave
// (= 4 queries); 183 M VALU wave-instructions per launch against 256 per cycle on the chip = 0.3 ms of the 0.41 ms: the kernel is
// bound by the ring bookkeeping on the vector ALUs, not by memory (L2 hit rate 95 %, 6.9 M L2 requests).  Skipping the cells whose
// nearest edge is already farther than the best candidate took 499 -> 413 us.
// (Measured dead end, round 3: ONE lane per query for the first rings -- bookkeeping paid once per 64 queries -- is 5x SLOWER, 2.6 ms per
// 720k queries: where the cloud is dense (ground returns next to the sensor) a query has hundreds of candidates in its 9-25 cells, and
// a lane walks them alone, one scattered 16-B row per step.)
__global__ __launch_bounds__(256) void knn_query_kernel(Level fine, Level coarse, int has_coarse, int n_ref,
                                                        const float* __restrict__ query, int qstride, int nq,
                                                        long long* __restrict__ index, float* __restrict__ dist_sqr,
                                                        int fine_max_rings) {
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup;  // query handled by this 16-lane group
    const int sub = threadIdx.x & (kGroup - 1);
    const int group_shift = (threadIdx.x & 63) & ~(kGroup - 1);  // first lane of this group inside the wavefront
    if (gid >= nq) return;
    const float qx = query[(size_t)gid * qstride], qy = query[(size_t)gid * qstride + 1], qz = query[(size_t)gid * qstride + 2];
    if (!(isfinite(qx) && isfinite(qy) && isfinite(qz)) || n_ref == 0) {
        if (sub == 0) { index[gid] = 0; if (dist_sqr) dist_sqr[gid] = nanf(""); }
        return;
    }
    float best = INFINITY;
    int best_i = 0x7fffffff;
    if (!ring_search(fine, qx, qy, qz, has_coarse ? fine_max_rings : -1, sub, group_shift, best, best_i))
        ring_search(coarse, qx, qy, qz, -1, sub, group_shift, best, best_i);
    if (sub != 0) return;
    index[gid] = best_i == 0x7fffffff ? 0 : best_i;
    if (dist_sqr) dist_sqr[gid] = best;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }
inline bool grid_ok(const liso_knn_grid* g) {
    return g && g->cell > 0.f && g->z_cell > 0.f && g->nx >= 1 && g->ny >= 1 && g->nz >= 1 &&
           (long)g->nx * g->ny * g->nz <= (1L << 24);
}
inline int n_keys(const liso_knn_grid* g) { return g->nx * g->ny * g->nz; }
// workspace: count[keys] | cursor[keys] | start[keys+1] | block_tot[kTotSlots] | key_of_pt[n] | (pad to 16 B) bucketed[n] float4
constexpr int kTotSlots = (1 << 14) + 2;  // 2^24 keys / 1024 per scan block, + grand total
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace

extern "C" {

size_t liso_knn_workspace_bytes(const liso_knn_grid* grid, int n_ref) {
    if (!grid_ok(grid) || n_ref < 0) return 0;
    const size_t cells = (size_t)n_keys(grid);
    return align16((3 * cells + 1 + kTotSlots + (size_t)n_ref) * sizeof(int)) + (size_t)n_ref * sizeof(float4) + 16;
}

int liso_knn_build_f32(const liso_knn_grid* grid, const float* ref, int ref_stride, int n_ref, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!grid_ok(grid) || n_ref < 0 || ref_stride < 3 || !workspace || (n_ref > 0 && !ref)) return LISO_EINVAL;
    if (workspace_bytes < liso_knn_workspace_bytes(grid, n_ref)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int cells = n_keys(grid);
    int* count = (int*)workspace;
    int* cursor = count + cells;
    int* start = cursor + cells;
    int* block_tot = start + cells + 1;
    int* cell_of_pt = block_tot + kTotSlots;
    float4* bucketed = (float4*)((char*)workspace + align16((3 * (size_t)cells + 1 + kTotSlots + (size_t)n_ref) * sizeof(int)));
    if (liso_zero::zero_async(count, 2 * (size_t)cells * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (n_ref > 0) knn_count_kernel<<<(n_ref + 255) / 256, 256, 0, st>>>(*grid, ref, ref_stride, n_ref, count, cell_of_pt);
    const int nsb = (cells + 1023) / 1024;
    knn_scan_block_kernel<<<nsb, 1024, 0, st>>>(count, cells, start, block_tot);
    knn_scan_tot_kernel<<<1, 1024, 0, st>>>(block_tot, nsb);
    knn_scan_add_kernel<<<nsb, 1024, 0, st>>>(start, cells, block_tot, nsb);
    if (n_ref > 0) knn_fill_kernel<<<(n_ref + 255) / 256, 256, 0, st>>>(ref, ref_stride, n_ref, cell_of_pt, start, cursor, bucketed);
    knn_occupancy_kernel<<<(grid->nx * grid->ny + 255) / 256, 256, 0, st>>>(*grid, start, (unsigned*)count);
    return check_launch();
}

static Level make_level(const liso_knn_grid* grid, const void* workspace, int n_ref) {
    Level l;
    l.g = *grid;
    const int cells = n_keys(grid);
    l.start = (const int*)workspace + 2 * (size_t)cells;
    l.occ = (const unsigned*)workspace;
    l.bucketed = (const float4*)((const char*)workspace + align16((3 * (size_t)cells + 1 + kTotSlots + (size_t)n_ref) * sizeof(int)));
    return l;
}

int liso_knn_sorted_ids(const liso_knn_grid* grid, const void* workspace, int n_ref, int64_t* ids, void* stream) {
    if (!grid_ok(grid) || n_ref < 0 || !workspace || (n_ref > 0 && !ids)) return LISO_EINVAL;
    if (n_ref == 0) return LISO_OK;
    const Level l = make_level(grid, workspace, n_ref);
    knn_sorted_ids_kernel<<<(n_ref + 255) / 256, 256, 0, (hipStream_t)stream>>>(l.bucketed, n_ref, (long long*)ids);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

int liso_knn_query_f32(const liso_knn_grid* grid, const void* workspace, const liso_knn_grid* coarse_grid,
                       const void* coarse_workspace, int n_ref, const float* query, int query_stride, int n_query,
                       int64_t* index, float* dist_sqr, int fine_max_rings, void* stream) {
    if (!grid_ok(grid) || n_ref < 0 || n_query < 0 || query_stride < 3 || !workspace) return LISO_EINVAL;
    if ((coarse_grid != nullptr) != (coarse_workspace != nullptr) || (coarse_grid && !grid_ok(coarse_grid))) return LISO_EINVAL;
    if (coarse_grid && fine_max_rings < 0) return LISO_EINVAL;
    if (n_query == 0) return LISO_OK;
    if (!query || !index) return LISO_EINVAL;
    const Level fine = make_level(grid, workspace, n_ref);
    const Level coarse = coarse_grid ? make_level(coarse_grid, coarse_workspace, n_ref) : fine;
    knn_query_kernel<<<(unsigned)(((long)n_query * kGroup + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        fine, coarse, coarse_grid != nullptr, n_ref, query, query_stride, n_query, (long long*)index, dist_sqr, fine_max_rings);
    return check_launch();
}

}  // extern "C"
