// DBSCAN over the dynamic BEV pillars + region moments, on the device, for gfx950.  C ABI + reference lines:
// include/liso_flow_cluster.h.
//
// The reference clusters the M dynamic pillars in the 5-D space (x, y, 2 fx, 2 fy, 2 fz) with sklearn's DBSCAN
// (eps 1 m, min_samples 5) on the host.  eps bounds the BEV distance, so every eps-neighbour of a pillar lies inside a
// (2R+1)^2 window of the dense grid (R = 6 at 0.195 m pillars): no neighbour lists, no KD-tree, no M^2 pairs.
//
//   dbscan_core     one thread per cell: counts eps-neighbours in the window (self included) -> core flag
//   dbscan_union    one thread per core cell: lock-free union-find (always link the larger root under the smaller:
//                   atomicMin on the parent) with every core eps-neighbour of smaller index
//   dbscan_flatten  one thread per core cell: is_root flag (the root is the smallest cell index of its component)
//   dbscan_label    one thread per dynamic cell: core -> rank of its root; border -> smallest rank among its core
//                   eps-neighbours; no core neighbour -> 0 (noise).  rank = 1-based position of the root among all
//                   roots in row-major order (an inclusive scan of the root flags, done by the caller).
// Why this equals sklearn's labelling: sklearn visits points in input (row-major) order, starts cluster k at the k-th
// still unlabelled core point and grows it completely (depth-first over core points, labelling every neighbour) before
// it moves on -- so clusters are the connected components of the core graph numbered by their smallest member, and a
// border point reachable from several clusters keeps the first one that reached it, i.e. the smallest number.
//
//   region_moments  one thread per labelled cell: 6 exact integer moments (n, sum r, sum c, sum r^2, sum c^2, sum rc)
//                   by 64-bit integer atomics (order independent)
//   region_props    one thread per label: centroid, orientation, axis lengths from the second central moments
//                   (skimage.measure.regionprops formulas, fp64)
#include <hip/hip_runtime.h>
#include "zero_fill.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liso_flow_cluster.h"
#include "../../include/liso_iou3d.h"

namespace {

struct Cfg {
    int batch, gx, gy, win, min_samples;
    double eps_sqr;
    float flow_weight;
};

// squared 5-D distance exactly as the host code builds it: float32 features (centre coordinates, flow_weight * flow)
// promoted to float64, differences squared and summed in feature order
__device__ __forceinline__ double dist_sqr(const float* __restrict__ xs, const float* __restrict__ ys,
                                           const float* __restrict__ flow, float fw, int r0, int c0, size_t i0, int r1, int c1,
                                           size_t i1) {
    const double d0 = (double)xs[r0] - (double)xs[r1];
    const double d1 = (double)ys[c0] - (double)ys[c1];
    const double d2 = (double)(fw * flow[3 * i0 + 0]) - (double)(fw * flow[3 * i1 + 0]);
    const double d3 = (double)(fw * flow[3 * i0 + 1]) - (double)(fw * flow[3 * i1 + 1]);
    const double d4 = (double)(fw * flow[3 * i0 + 2]) - (double)(fw * flow[3 * i1 + 2]);
    return d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 + d4 * d4;
}

template <typename F>
__device__ __forceinline__ void for_each_neighbour(const Cfg& c, const uint8_t* __restrict__ dyn, const float* __restrict__ xs,
                                                   const float* __restrict__ ys, const float* __restrict__ flow, int b, int r,
                                                   int col, F&& f) {
    const size_t base = (size_t)b * c.gx * c.gy;
    const size_t me = base + (size_t)r * c.gy + col;
    const int r_lo = max(r - c.win, 0), r_hi = min(r + c.win, c.gx - 1);
    const int c_lo = max(col - c.win, 0), c_hi = min(col + c.win, c.gy - 1);
    for (int rr = r_lo; rr <= r_hi; rr++)
        for (int cc = c_lo; cc <= c_hi; cc++) {
            const size_t nb = base + (size_t)rr * c.gy + cc;
            if (!dyn[nb]) continue;
            if (dist_sqr(xs, ys, flow, c.flow_weight, r, col, me, rr, cc, nb) <= c.eps_sqr) f(nb);
        }
}

// Wave-cooperative window walk.  Dynamic pillars are sparse (a few thousand of 512 x 512) and come in blobs a handful of
// columns wide, so a wave of 64 consecutive cells holds 0-10 of them: instead of every such lane walking its 13 x 13 window
// alone (169 dependent iterations with 1-10 of 64 lanes busy), the wave takes its flagged cells one after the other and all
// 64 lanes share the window of that cell (3 iterations, consecutive columns = coalesced reads).  f(nb, leader_lane) runs on the
// lanes whose window cell is a neighbour (dynamic, within eps) of the flagged cell.
constexpr int kDenseWave = 20;  // flagged cells per wave above which the per-lane walk (169 iterations, all lanes busy) is cheaper

template <typename F>
__device__ __forceinline__ void wave_for_each_neighbour(const Cfg& c, const uint8_t* __restrict__ dyn, const float* __restrict__ xs,
                                                        const float* __restrict__ ys, const float* __restrict__ flow, size_t me,
                                                        int lane, F&& f) {
    const size_t per = (size_t)c.gx * c.gy;
    const int b = (int)(me / per), r = (int)((me % per) / c.gy), col = (int)(me % c.gy);
    const size_t base = (size_t)b * per;
    const int r_lo = max(r - c.win, 0), r_hi = min(r + c.win, c.gx - 1);
    const int c_lo = max(col - c.win, 0), c_hi = min(col + c.win, c.gy - 1);
    const int wc = c_hi - c_lo + 1, n = (r_hi - r_lo + 1) * wc;
    for (int k = lane; k < n; k += 64) {
        const int rr = r_lo + k / wc, cc = c_lo + k % wc;
        const size_t nb = base + (size_t)rr * c.gy + cc;
        if (dyn[nb] && dist_sqr(xs, ys, flow, c.flow_weight, r, col, me, rr, cc, nb) <= c.eps_sqr) f(nb);
    }
}

__global__ __launch_bounds__(256) void dbscan_core_kernel(Cfg c, const uint8_t* __restrict__ dyn, const float* __restrict__ xs,
                                                          const float* __restrict__ ys, const float* __restrict__ flow,
                                                          uint8_t* __restrict__ core, int* __restrict__ parent) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)c.gx * c.gy, total = per * c.batch;
    const int lane = threadIdx.x & 63;
    const bool flagged = i < total && dyn[i];
    unsigned long long todo = __ballot(flagged);
    int is_core = 0;
    if (__popcll(todo) > kDenseWave) {  // a wave inside a large dynamic region: every lane walks its own window (lanes all busy)
        if (flagged) {
            int count = 0;
            for_each_neighbour(c, dyn, xs, ys, flow, (int)(i / per), (int)((i % per) / c.gy), (int)(i % c.gy), [&](size_t) { count++; });
            is_core = count >= c.min_samples;
        }
        todo = 0ull;
    }
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        int count = 0;
        wave_for_each_neighbour(c, dyn, xs, ys, flow, i - lane + src, lane, [&](size_t) { count++; });
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) count += __shfl_xor(count, o);
        if (lane == src) is_core = count >= c.min_samples;
    }
    if (i >= total) return;
    core[i] = (uint8_t)is_core;
    parent[i] = is_core ? (int)(i % per) : -1;  // per-sample cell index
}

// parents only ever move to smaller indices; device-scope loads keep concurrent hooks visible (fewer retries)
__device__ __forceinline__ int find_root(const int* parent, int x) {
    int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != x) { x = p; p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    return x;
}

// find + one-step compression: the start node is re-parented to the root that was found (an ancestor with a smaller index:
// parents still only ever decrease, so concurrent hooks stay valid); the next walk from this node is one hop
__device__ __forceinline__ int find_root_compress(int* parent, int x0) {
    const int r = find_root(parent, x0);
    if (r != x0) atomicMin(&parent[x0], r);
    return r;
}

__global__ __launch_bounds__(256) void dbscan_union_kernel(Cfg c, const uint8_t* __restrict__ dyn, const uint8_t* __restrict__ core,
                                                           const float* __restrict__ xs, const float* __restrict__ ys,
                                                           const float* __restrict__ flow, int* __restrict__ parent) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)c.gx * c.gy, total = per * c.batch;
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(i < total && core[i]);
    if (__popcll(todo) > kDenseWave) {
        if (i < total && core[i]) {
            const int b = (int)(i / per), r = (int)((i % per) / c.gy), col = (int)(i % c.gy);
            int* par = parent + (size_t)b * per;
            const int me = (int)(i % per);
            for_each_neighbour(c, dyn, xs, ys, flow, b, r, col, [&](size_t nb) {
                if (!core[nb]) return;
                const int other = (int)(nb % per);
                if (other >= me) return;
                int x = me, y = other;
                while (true) {
                    x = find_root_compress(par, x);
                    y = find_root_compress(par, y);
                    if (x == y) break;
                    if (x < y) { const int t = x; x = y; y = t; }
                    const int old = atomicMin(&par[x], y);
                    if (old == x) break;
                    x = old;
                }
            });
        }
        return;
    }
    while (todo) {  // (the final forest -- every tree rooted at its smallest cell -- does not depend on the order of the hooks)
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const size_t cell = i - lane + src;
        int* par = parent + (cell / per) * per;
        const int me = (int)(cell % per);
        wave_for_each_neighbour(c, dyn, xs, ys, flow, cell, lane, [&](size_t nb) {
            if (!core[nb]) return;
            const int other = (int)(nb % per);
            if (other >= me) return;  // every undirected edge once
            int x = me, y = other;
            while (true) {
                x = find_root_compress(par, x);
                y = find_root_compress(par, y);
                if (x == y) break;
                if (x < y) { const int t = x; x = y; y = t; }  // x: larger root, linked under y
                const int old = atomicMin(&par[x], y);
                if (old == x) break;  // x was still a root: linked
                x = old;              // somebody re-parented x meanwhile: retry from there
            }
        });
    }
}

// ---- union-find in two levels: LDS inside a tile (+ halo), global atomics for one edge per cell ----------------------------------------
// dbscan_union_kernel hooks every (core cell, smaller core eps-neighbour) pair with device-scope atomics on the global parent array: a
// blob of 150 cells is ~10 000 hooks, each a chain of dependent L2 round trips (round-5 measurement on the bench's sweeps: 4 000 core
// cells, 107 us).  Here a block owns a kTR x kTC tile: the core flags and flows of the tile and its eps halo go to LDS, every edge
// (x in the tile, y in x's window, y < x) is hooked in an LDS parent array (values = per-sample cell indices, so local roots are the
// smallest cell of the local component, the order the global forest uses), and then every core cell of the region is hooked ONCE in
// the global array: with its local root.  Each eps edge lies inside the region of the tile that holds its larger end, so the global
// forest connects exactly the components of the core graph; tiles without a core cell (almost all of them) leave after one load.
constexpr int kTR = 16, kTC = 32, kUfThreads = 256, kMaxWin = 8;
constexpr int kRegMax = (kTR + 2 * kMaxWin) * (kTC + 2 * kMaxWin);

__device__ __forceinline__ int lds_find(const int* par, int l, int r0, int c0, int RW, int gy) {
    // par[l] = per-sample cell index of the parent; a root points at itself
    while (true) {
        const int me = (r0 + l / RW) * gy + (c0 + l % RW);
        const int p = par[l];
        if (p == me) return l;
        l = (p / gy - r0) * RW + (p % gy - c0);
    }
}

__global__ __launch_bounds__(kUfThreads) void dbscan_union_tiled_kernel(Cfg c, const uint8_t* __restrict__ dyn,
                                                                        const uint8_t* __restrict__ core, const float* __restrict__ xs,
                                                                        const float* __restrict__ ys, const float* __restrict__ flow,
                                                                        int* __restrict__ parent, int tiles_r, int tiles_c) {
    __shared__ int s_par[kRegMax];
    __shared__ float s_f[kRegMax][3];
    __shared__ uint8_t s_core[kRegMax];
    __shared__ int s_any;
    const int b = blockIdx.y;
    const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int tid = threadIdx.x;
    const size_t per = (size_t)c.gx * c.gy, base = (size_t)b * per;
    // does the TILE hold a core cell?
    if (tid == 0) s_any = 0;
    __syncthreads();
    int any = 0;
    for (int q = tid; q < kTR * kTC; q += kUfThreads) {
        const int r = tr * kTR + q / kTC, col = tc * kTC + q % kTC;
        if (r < c.gx && col < c.gy && core[base + (size_t)r * c.gy + col]) any = 1;
    }
    if (any) s_any = 1;
    __syncthreads();
    if (!s_any) return;
    // region = tile + halo, clipped to the grid
    const int r0 = max(tr * kTR - c.win, 0), r1 = min(tr * kTR + kTR + c.win, c.gx);
    const int c0 = max(tc * kTC - c.win, 0), c1 = min(tc * kTC + kTC + c.win, c.gy);
    const int RW = c1 - c0, RH = r1 - r0, RN = RW * RH;
    for (int l = tid; l < RN; l += kUfThreads) {
        const int r = r0 + l / RW, col = c0 + l % RW;
        const size_t g = base + (size_t)r * c.gy + col;
        const uint8_t co = core[g];
        s_core[l] = co;
        s_par[l] = r * c.gy + col;
        if (co) {
            s_f[l][0] = c.flow_weight * flow[3 * g + 0];
            s_f[l][1] = c.flow_weight * flow[3 * g + 1];
            s_f[l][2] = c.flow_weight * flow[3 * g + 2];
        }
    }
    __syncthreads();
    // local hooks: every core cell of the tile with its smaller core eps-neighbours.  The tile's core cells are listed first; a wave takes
    // a cell at a time and its 64 lanes share the cell's window (the half with smaller indices: <= (2 win + 1) win + win cells)
    __shared__ int s_list[kTR * kTC];
    __shared__ int s_n;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int q = tid; q < kTR * kTC; q += kUfThreads) {
        const int r = tr * kTR + q / kTC, col = tc * kTC + q % kTC;
        if (r < c.gx && col < c.gy && s_core[(r - r0) * RW + (col - c0)]) s_list[atomicAdd(&s_n, 1)] = q;
    }
    __syncthreads();
    const int n_list = s_n, lane = tid & 63, wave = tid >> 6;
    const int ww = 2 * c.win + 1;
    for (int e = wave; e < n_list; e += kUfThreads / 64) {
        const int q = s_list[e];
        const int r = tr * kTR + q / kTC, col = tc * kTC + q % kTC;
        const int lx = (r - r0) * RW + (col - c0);
        const double x0 = (double)xs[r], y0 = (double)ys[col];
        const double f0 = (double)s_f[lx][0], f1 = (double)s_f[lx][1], f2 = (double)s_f[lx][2];
        const int n_win = c.win * ww + c.win;  // window cells in front of the centre, row-major
        for (int k = lane; k < n_win; k += 64) {
            const int rr = r - c.win + k / ww, cc = col - c.win + k % ww;
            if (rr < 0 || cc < 0 || cc >= c.gy) continue;  // (rr <= r < gx)
            const int ly = (rr - r0) * RW + (cc - c0);
            if (!s_core[ly]) continue;
            // the 5-D distance exactly as dist_sqr builds it (fp32 features promoted to fp64, summed in feature order)
            const double d0 = x0 - (double)xs[rr], d1 = y0 - (double)ys[cc];
            const double d2 = f0 - (double)s_f[ly][0], d3 = f1 - (double)s_f[ly][1], d4 = f2 - (double)s_f[ly][2];
            if (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 + d4 * d4 > c.eps_sqr) continue;
            int a = lx, bq = ly;
            while (true) {
                a = lds_find(s_par, a, r0, c0, RW, c.gy);
                bq = lds_find(s_par, bq, r0, c0, RW, c.gy);
                if (a == bq) break;
                const int ga = (r0 + a / RW) * c.gy + (c0 + a % RW), gb = (r0 + bq / RW) * c.gy + (c0 + bq % RW);
                int hi = a, lo_g = gb, hi_g = ga;
                if (ga < gb) { hi = bq; lo_g = ga; hi_g = gb; }
                const int old = atomicMin(&s_par[hi], lo_g);  // the larger root under the smaller one
                if (old == hi_g) break;                       // it was still a root: linked
                // somebody re-parented it meanwhile, to `old`.  The atomicMin has already stored min(old, lo_g) there, so when
                // lo_g < old the edge hi -> old is GONE: go on with the pair (old, lo_g), as the global loops do with `x = old`
                // (walking on from hi would find hi and lo_g joined and leave old's tree cut off)
                a = (old / c.gy - r0) * RW + (old % c.gy - c0);
                bq = (lo_g / c.gy - r0) * RW + (lo_g % c.gy - c0);
            }
        }
    }
    __syncthreads();
    // one global hook per core cell of the region: the cell with its local root
    int* par = parent + base;
    for (int l = tid; l < RN; l += kUfThreads) {
        if (!s_core[l]) continue;
        const int me = (r0 + l / RW) * c.gy + (c0 + l % RW);
        const int lr = lds_find(s_par, l, r0, c0, RW, c.gy);
        const int root = (r0 + lr / RW) * c.gy + (c0 + lr % RW);
        if (root == me) continue;
        int x = me, y = root;
        while (true) {
            x = find_root_compress(par, x);
            y = find_root_compress(par, y);
            if (x == y) break;
            if (x < y) { const int t = x; x = y; y = t; }
            const int old = atomicMin(&par[x], y);
            if (old == x) break;
            x = old;
        }
    }
}

__global__ void dbscan_flatten_kernel(Cfg c, const uint8_t* __restrict__ core, int* __restrict__ parent,
                                      int* __restrict__ is_root) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)c.gx * c.gy;
    if (i >= per * c.batch) return;
    int root_flag = 0;
    if (core[i]) {
        const int* par = parent + (i / per) * per;
        const int root = find_root(par, (int)(i % per));
        root_flag = root == (int)(i % per);  // no path compression: the label pass walks the (short) chains again
    }
    is_root[i] = root_flag;
}

__global__ void dbscan_label_kernel(Cfg c, const uint8_t* __restrict__ dyn, const uint8_t* __restrict__ core,
                                    const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ flow,
                                    const int* __restrict__ parent, const int* __restrict__ root_rank, int* __restrict__ labels) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)c.gx * c.gy;
    if (i >= per * c.batch) return;
    int label = 0;
    if (dyn[i]) {
        const int b = (int)(i / per), r = (int)((i % per) / c.gy), col = (int)(i % c.gy);
        const int* par = parent + (size_t)b * per;
        const int* rank = root_rank + (size_t)b * per;
        if (core[i]) {
            label = rank[find_root(par, (int)(i % per))];
        } else {
            int best = 0x7fffffff;
            for_each_neighbour(c, dyn, xs, ys, flow, b, r, col, [&](size_t nb) {
                if (core[nb]) best = min(best, rank[find_root(par, (int)(nb % per))]);
            });
            label = best == 0x7fffffff ? 0 : best;
        }
    }
    labels[i] = label;
}

// Integer moments of every labelled region.  A wave covers 64 consecutive cells, which carry at most a few distinct labels:
// the wave reduces the six sums per label (butterfly) and ONE lane issues the atomics -- 6 per (wave, label) instead of 6 per
// cell, which serialised on the few dozen label rows (243 us at 512 x 512).  Integer sums: any order gives the same bits.
__global__ __launch_bounds__(256) void region_moments_kernel(const int* __restrict__ labels, int batch, int gx, int gy, int max_labels,
                                                             unsigned long long* __restrict__ mom) {
    // (32-bit index arithmetic: batch * gx * gy < 2^31 is checked by the caller; three 64-bit divisions per cell were most of this
    // kernel's instructions)
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned per = (unsigned)gx * (unsigned)gy;
    const int lane = threadIdx.x & 63;
    int l = i < per * (unsigned)batch ? labels[i] : 0;
    if (l > max_labels) l = 0;
    const unsigned bi = i / per, rem = i - bi * per, ri = rem / (unsigned)gy;
    const long key = l > 0 ? (long)bi * max_labels + (l - 1) : -1;
    const unsigned long long r = ri, col = rem - ri * (unsigned)gy;
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        const long k = __shfl(key, src);
        const bool mine = key == k;
        unsigned long long v[6] = {mine ? 1ull : 0ull, mine ? r : 0ull, mine ? col : 0ull, mine ? r * r : 0ull, mine ? col * col : 0ull,
                                   mine ? r * col : 0ull};
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int q = 0; q < 6; q++) v[q] += __shfl_xor(v[q], o);
        if (lane == src) {
            unsigned long long* m = mom + (size_t)k * 6;
#pragma unroll
            for (int q = 0; q < 6; q++) atomicAdd(m + q, v[q]);
        }
        todo &= ~__ballot(mine);
    }
}

// The same sums without global atomics, for label maps whose labelled cells are SCATTERED (a frozen flow network on real data gives
// blobs; an untrained one marks single pillars all over the map): round-5 measurement on the bench's sweeps -- 4 306 labelled cells in
// 30 clusters, nearly every one in a wave of its own -> 26 k 64-bit atomics on 180 addresses, 75 us; 30 compact blobs of the same
// area: 4 us.  Here a block owns kRmCells consecutive cells of one sample, adds into LDS ([max_labels][6] u64, ds_add_u64) and writes
// its rows to partial[sample][block][label][6] (plain stores, zeros included: no fill pass); region_props_kernel adds the blocks'
// rows (integers: any order gives the same bits) before it evaluates the moments.
constexpr int kRmCells = 2048, kRmThreads = 256, kRmMaxLabels = 1024;
__global__ __launch_bounds__(kRmThreads) void region_moments_lds_kernel(const int* __restrict__ labels, int gx, int gy, int max_labels,
                                                                         int blocks_per_sample, unsigned long long* __restrict__ partial) {
    extern __shared__ unsigned long long s_mom[];  // [max_labels][6]
    const int b = blockIdx.y, blk = blockIdx.x;
    const unsigned per = (unsigned)gx * (unsigned)gy;
    for (int q = threadIdx.x; q < max_labels * 6; q += kRmThreads) s_mom[q] = 0ull;
    __syncthreads();
    const unsigned c0 = (unsigned)blk * kRmCells;
    for (unsigned c = c0 + threadIdx.x; c < c0 + kRmCells && c < per; c += kRmThreads) {
        int l = labels[(size_t)b * per + c];
        if (l <= 0 || l > max_labels) continue;
        const unsigned long long r = c / (unsigned)gy, col = c - (unsigned)r * (unsigned)gy;
        unsigned long long* m = s_mom + (size_t)(l - 1) * 6;
        atomicAdd(m + 0, 1ull); atomicAdd(m + 1, r); atomicAdd(m + 2, col);
        atomicAdd(m + 3, r * r); atomicAdd(m + 4, col * col); atomicAdd(m + 5, r * col);
    }
    __syncthreads();
    unsigned long long* out = partial + ((size_t)b * blocks_per_sample + blk) * max_labels * 6;
    for (int q = threadIdx.x; q < max_labels * 6; q += kRmThreads) out[q] = s_mom[q];
}

// mom[region][6] = sum over the sample's blocks of partial[sample][block][label][6]: one wave per region, lanes stride over the blocks
__global__ __launch_bounds__(256) void region_moments_sum_kernel(const unsigned long long* __restrict__ partial, int batch, int max_labels,
                                                                 int blocks_per_sample, unsigned long long* __restrict__ mom) {
    const int region = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (region >= batch * max_labels) return;
    const int b = region / max_labels, l = region - b * max_labels;
    unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
    for (int blk = lane; blk < blocks_per_sample; blk += 64) {
        const unsigned long long* p = partial + (((size_t)b * blocks_per_sample + blk) * max_labels + l) * 6;
#pragma unroll
        for (int q = 0; q < 6; q++) v[q] += p[q];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < 6; q++) v[q] += __shfl_xor(v[q], o);
    if (lane < 6) {
        unsigned long long w = v[0];
#pragma unroll
        for (int q = 1; q < 6; q++) w = lane == q ? v[q] : w;
        mom[(size_t)region * 6 + lane] = w;
    }
}

__global__ void region_props_kernel(const unsigned long long* __restrict__ mom, long n_regions, double* __restrict__ props) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_regions) return;
    const unsigned long long* m = mom + (size_t)i * 6;
    double* p = props + (size_t)i * 5;
    const double n = (double)m[0];
    if (m[0] == 0) { p[0] = p[1] = p[2] = p[3] = p[4] = 0.0; return; }
    const double sr = (double)m[1], sc = (double)m[2];
    const double cr = sr / n, cc = sc / n;  // centroid (row, col)
    // central second moments, normalised by the area (skimage: inertia_tensor = [[mu02, -mu11], [-mu11, mu20]] / mu00)
    const double mu20 = ((double)m[3] - sr * sr / n) / n;  // rows
    const double mu02 = ((double)m[4] - sc * sc / n) / n;  // cols
    const double mu11 = ((double)m[5] - sr * sc / n) / n;
    const double a = mu02, b = -mu11, c = mu20;
    double orientation;
    if (a - c == 0.0)
        orientation = b < 0.0 ? -M_PI / 4.0 : M_PI / 4.0;
    else
        orientation = 0.5 * atan2(-2.0 * b, c - a);
    const double tr = a + c, disc = sqrt((a - c) * (a - c) + 4.0 * b * b);
    const double l1 = fmax(0.5 * (tr + disc), 0.0), l2 = fmax(0.5 * (tr - disc), 0.0);
    p[0] = cr; p[1] = cc; p[2] = orientation; p[3] = 4.0 * sqrt(l1); p[4] = 4.0 * sqrt(l2);
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }
inline bool cfg_ok(const liso_dbscan_cfg* c) {
    return c && c->batch >= 1 && c->gx >= 1 && c->gy >= 1 && c->window >= 0 && c->min_samples >= 1 && c->eps > 0.f &&
           (long)c->gx * c->gy < (1L << 31);
}
inline Cfg to_cfg(const liso_dbscan_cfg* c) {
    Cfg k;
    k.batch = c->batch; k.gx = c->gx; k.gy = c->gy; k.win = c->window; k.min_samples = c->min_samples;
    k.eps_sqr = (double)c->eps * (double)c->eps;
    k.flow_weight = c->flow_weight;
    return k;
}

}  // namespace

extern "C" {

int liso_dbscan_components(const liso_dbscan_cfg* cfg, const uint8_t* dynamic_mask, const float* row_coords,
                           const float* col_coords, const float* flow, uint8_t* core, int32_t* parent, int32_t* is_root,
                           void* stream) {
    if (!cfg_ok(cfg) || !dynamic_mask || !row_coords || !col_coords || !flow || !core || !parent || !is_root) return LISO_EINVAL;
    const Cfg c = to_cfg(cfg);
    const size_t total = (size_t)c.batch * c.gx * c.gy;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    dbscan_core_kernel<<<blocks, 256, 0, st>>>(c, dynamic_mask, row_coords, col_coords, flow, core, parent);
    static const bool flat_union = getenv("LISO_DBSCAN_FLAT_UNION") != nullptr && atoi(getenv("LISO_DBSCAN_FLAT_UNION")) != 0;
    if (c.win <= kMaxWin && !flat_union) {
        const int tiles_r = (c.gx + kTR - 1) / kTR, tiles_c = (c.gy + kTC - 1) / kTC;
        dbscan_union_tiled_kernel<<<dim3(tiles_r * tiles_c, c.batch), kUfThreads, 0, st>>>(c, dynamic_mask, core, row_coords, col_coords,
                                                                                         flow, parent, tiles_r, tiles_c);
    } else {
        dbscan_union_kernel<<<blocks, 256, 0, st>>>(c, dynamic_mask, core, row_coords, col_coords, flow, parent);
    }
    dbscan_flatten_kernel<<<blocks, 256, 0, st>>>(c, core, parent, is_root);
    return check_launch();
}

int liso_dbscan_labels(const liso_dbscan_cfg* cfg, const uint8_t* dynamic_mask, const float* row_coords, const float* col_coords,
                       const float* flow, const uint8_t* core, const int32_t* parent, const int32_t* root_rank,
                       int32_t* labels, void* stream) {
    if (!cfg_ok(cfg) || !dynamic_mask || !row_coords || !col_coords || !flow || !core || !parent || !root_rank || !labels)
        return LISO_EINVAL;
    const Cfg c = to_cfg(cfg);
    const size_t total = (size_t)c.batch * c.gx * c.gy;
    dbscan_label_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(c, dynamic_mask, core, row_coords,
                                                                                       col_coords, flow, parent, root_rank, labels);
    return check_launch();
}

int liso_region_props(const int32_t* labels, int batch, int gx, int gy, int max_labels, uint64_t* moments, double* props,
                      void* stream) {
    if (!labels || batch < 1 || gx < 1 || gy < 1 || max_labels < 1 || !moments || !props) return LISO_EINVAL;
    if ((size_t)batch * gx * gy >= (1ull << 31)) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)batch * gx * gy;
    const long regions = (long)batch * max_labels;
    if (liso_zero::zero_async(moments, (size_t)regions * 6 * sizeof(uint64_t), st) != hipSuccess) return LISO_ELAUNCH;
    region_moments_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(labels, batch, gx, gy, max_labels,
                                                                          (unsigned long long*)moments);
    region_props_kernel<<<(unsigned)((regions + 255) / 256), 256, 0, st>>>((const unsigned long long*)moments, regions, props);
    return check_launch();
}

size_t liso_region_props_workspace_bytes(int batch, int gx, int gy, int max_labels) {
    if (batch < 1 || gx < 1 || gy < 1 || max_labels < 1 || max_labels > kRmMaxLabels) return 0;
    const size_t per = (size_t)gx * gy;
    if (per * batch >= (1ull << 31)) return 0;
    const size_t blocks = (per + kRmCells - 1) / kRmCells;
    return (size_t)batch * blocks * max_labels * 6 * sizeof(uint64_t);
}

int liso_region_props_ws(const int32_t* labels, int batch, int gx, int gy, int max_labels, uint64_t* moments, double* props,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!labels || batch < 1 || gx < 1 || gy < 1 || max_labels < 1 || !moments || !props) return LISO_EINVAL;
    const size_t need = liso_region_props_workspace_bytes(batch, gx, gy, max_labels);
    if (need == 0) return liso_region_props(labels, batch, gx, gy, max_labels, moments, props, stream);  // (> 1024 labels: atomics path)
    if (!workspace || workspace_bytes < need) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)(((size_t)gx * gy + kRmCells - 1) / kRmCells);
    const long regions = (long)batch * max_labels;
    region_moments_lds_kernel<<<dim3(blocks, batch), kRmThreads, (size_t)max_labels * 6 * sizeof(uint64_t), st>>>(
        labels, gx, gy, max_labels, blocks, (unsigned long long*)workspace);
    region_moments_sum_kernel<<<(unsigned)((regions + 3) / 4), 256, 0, st>>>((const unsigned long long*)workspace, batch, max_labels,
                                                                            blocks, (unsigned long long*)moments);
    region_props_kernel<<<(unsigned)((regions + 255) / 256), 256, 0, st>>>((const unsigned long long*)moments, regions, props);
    return check_launch();
}

}  // extern "C"
