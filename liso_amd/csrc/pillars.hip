// Pillar path for gfx950 (MI355X): hard voxelisation -> fused PillarFeatureNet -> dense BEV scatter (+ backward).
// C ABI and the reference lines each stage replaces: include/liso_pillars.h.
//
// Design (HBM-bound stage; nothing here is GEMM-shaped enough for MFMA: K = C+6 <= 11):
//   * voxelise is deterministic:
//       assign      per point: cell id, atomicAdd(count[cell]), atomicMax(first[cell], INT_MAX - i)
//       tile_count  per 1024-point tile: how many points are the first of their cell
//       rank        voxel ordinal = exclusive prefix of "is first" in point order (the reference's voxel order),
//                   voxels >= max_voxels dropped, coors / num_points / cell->row written
//       cell_scan / seg_fill / seg_place: every cell owns a segment of point indices (exclusive prefix of the counts); a point's
//                   slot is its arrival rank = the number of smaller indices in its cell's segment (no sort, no library call)
//       (an earlier atomicMin insertion cascade cost 290 us at B = 4: adjacent LiDAR rays hit the same pillar together)
//   * the PFN never materialises voxels[P,20,C], [P,20,10] or [P,20,64]: `pfn_decorate` walks count -> slot indices -> points once
//     (32 lanes per pillar) and writes 12-float feature rows in pillar order (CSR); statistics / forward / backward stream those
//     rows: a wave stages the rows of 4 consecutive pillars in LDS, the 64 lanes ARE the 64 output channels (running max in a
//     register, one coalesced 64-channel store per pillar into the channels-last canvas; empty cells are zero-filled by other
//     blocks of the same launch: every canvas cell is written exactly once, no memset pass).
//   * no hipMemsetAsync anywhere (zero_fill.h): memset nodes do not survive hipGraph replays on this runtime.  (Until round 4 rocPRIM's
//     radix sort, which does memset internally, kept this encoder in front of every graph: liso_amd/utils/graph_safety.py.)
//   * BatchNorm1d batch statistics come from the second moments of the 10-vector (sum f f^T in fp64, 66 numbers)
//     instead of 2x64 per-channel sums over [P*20, 64]; the same moments give the BN backward terms in closed form.
//   * all cross-workgroup reductions are two-stage with fixed order (no float atomics): results are reproducible.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "zero_fill.h"
#include <cstring>
#include <hip/hip_bf16.h>
#include <limits.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"  // error codes
#include "../../include/liso_pillars.h"

namespace {

constexpr int kTile = 1024;          // points per tile in the rank pass
constexpr int kOut = LISO_PFN_OUT;
constexpr int kFP = 12;              // padded feature row in LDS (F <= 11, +1 augmented "1")
constexpr int kMaxPts = 32;          // max_points supported by the one-wave-per-pillar mapping
constexpr int kPfnThreads = 256;
constexpr int kPfnGrid = 1024;       // persistent grid for stats/backward (fixed => deterministic partial order)

struct BatchInfo {
    int off[LISO_PILLARS_MAX_BATCH + 1];       // point offsets
    int tile_off[LISO_PILLARS_MAX_BATCH + 1];  // tile offsets
};

__device__ __forceinline__ int sample_of(const BatchInfo& bi, int batch, int i) {
    int b = 0;
    while (b + 1 < batch && i >= bi.off[b + 1]) b++;
    return b;
}

// ---------------------------------------------------------------------------------------------------------
// voxelise
// ---------------------------------------------------------------------------------------------------------
// voxel_generator.py:257-264: c = floor((p - range_min) / voxel_size) per axis, dropped if c < 0 or c >= grid
__global__ void assign_kernel(const float* __restrict__ pts, int n_total, int C, BatchInfo bi, int batch,
                              liso_pillar_cfg cfg, int* __restrict__ cell_of_point, int* __restrict__ count,
                              int* __restrict__ first_enc, int* __restrict__ pos_in_cell) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cell = -1;
    if (i < n_total) {
        const float* p = pts + (size_t)i * C;
        const float cx = floorf((p[0] - cfg.x_min) / cfg.vx);
        const float cy = floorf((p[1] - cfg.y_min) / cfg.vy);
        const float cz = floorf((p[2] - cfg.z_min) / cfg.vz);
        // NaN fails every comparison below and is dropped
        const bool ok = cx >= 0.f && cx < (float)cfg.gx && cy >= 0.f && cy < (float)cfg.gy && cz >= 0.f && cz < 1.f;
        if (ok) cell = (sample_of(bi, batch, i) * cfg.gx + (int)cx) * cfg.gy + (int)cy;
        cell_of_point[i] = cell;
    }
    // Adjacent LiDAR returns hit the same pillar: one pair of atomics per RUN of equal cells in the wave, all runs in ONE atomic
    // instruction (round 4 walked the distinct cells of the wave one after the other: 64 single-lane atomic instructions per wave on
    // clouds in random point order, 69 us per call).  The returning add gives every run a private range of its cell's segment:
    // pos_in_cell = range start + offset inside the run -- unique per point, which is all seg_fill_kernel needs (arrival order is
    // restored by seg_place_kernel from the indices themselves).
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(cell, 1);
    const unsigned long long heads = __ballot(lane == 0 || cell != prev);
    const unsigned long long below = heads & ((2ull << lane) - 1ull);        // heads at or below this lane (never empty: lane 0 is one)
    const int head_lane = 63 - __builtin_clzll(below);
    const unsigned long long above = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
    const int next_head = above ? __ffsll((long long)above) - 1 : 64;
    int base = 0;
    if (cell >= 0 && head_lane == lane) {
        base = atomicAdd(&count[cell], next_head - lane);  // (the run ends in front of the next head)
        atomicMax(&first_enc[cell], INT_MAX - i);          // == atomicMin over i with a zero-initialised array (the head is the run's smallest i)
    }
    base = __shfl(base, head_lane);
    if (i < n_total) pos_in_cell[i] = cell >= 0 ? base + (lane - head_lane) : 0;
}

__device__ __forceinline__ bool is_first(const int* cell_of_point, const int* first_enc, int i) {
    const int cell = cell_of_point[i];
    return cell >= 0 && first_enc[cell] == INT_MAX - i;
}

__global__ __launch_bounds__(kTile) void tile_count_kernel(const int* __restrict__ cell_of_point,
                                                           const int* __restrict__ first_enc, BatchInfo bi, int batch,
                                                           int* __restrict__ tile_count) {
    __shared__ int wsum[kTile / 64];
    const int tile = blockIdx.x;
    int b = 0;
    while (b + 1 < batch && tile >= bi.tile_off[b + 1]) b++;
    const int i = bi.off[b] + (tile - bi.tile_off[b]) * kTile + threadIdx.x;
    const bool f = i < bi.off[b + 1] && is_first(cell_of_point, first_enc, i);
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kTile / 64; w++) s += wsum[w];
        tile_count[tile] = s;
    }
}

// voxel_generator.py:265-279: new voxel on first sight (if voxel_num < max_voxels), coors recorded, count capped
__global__ __launch_bounds__(kTile) void rank_kernel(const int* __restrict__ cell_of_point,
                                                     const int* __restrict__ first_enc,
                                                     const int* __restrict__ count, BatchInfo bi, int batch,
                                                     liso_pillar_cfg cfg, const int* __restrict__ tile_count,
                                                     int* __restrict__ coors, int* __restrict__ num_points,
                                                     int* __restrict__ cell_to_voxel, int* __restrict__ num_voxels) {
    __shared__ int wsum[kTile / 64];
    __shared__ int base_s;
    const int tile = blockIdx.x;
    int b = 0;
    while (b + 1 < batch && tile >= bi.tile_off[b + 1]) b++;
    // ordinal base = first-points in the preceding tiles of this sample (one wave, fixed order)
    if (threadIdx.x < 64) {
        int s = 0;
        for (int t = bi.tile_off[b] + threadIdx.x; t < tile; t += 64) s += tile_count[t];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (threadIdx.x == 0) base_s = s;
    }
    const int i = bi.off[b] + (tile - bi.tile_off[b]) * kTile + threadIdx.x;
    const bool f = i < bi.off[b + 1] && is_first(cell_of_point, first_enc, i);
    const unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int pre = base_s;
    for (int w = 0; w < wave; w++) pre += wsum[w];
    if (f) {
        const int ord = pre + __popcll(m & ((1ULL << lane) - 1ULL));
        if (ord < cfg.max_voxels) {
            const int cell = cell_of_point[i];
            const int v = b * cfg.max_voxels + ord;
            cell_to_voxel[cell] = v + 1;
            const int within = cell - b * cfg.gx * cfg.gy;
            coors[v * 4 + 0] = b;
            coors[v * 4 + 1] = 0;
            coors[v * 4 + 2] = within / cfg.gy;  // x index
            coors[v * 4 + 3] = within % cfg.gy;  // y index
            const int c = count[cell];
            num_points[v] = c < cfg.max_points ? c : cfg.max_points;
        }
    }
    // last tile of the sample publishes the voxel count
    if (tile == bi.tile_off[b + 1] - 1 && threadIdx.x == 0) {
        int tot = base_s;
        for (int w = 0; w < kTile / 64; w++) tot += wsum[w];
        num_voxels[b] = tot < cfg.max_voxels ? tot : cfg.max_voxels;
    }
}

// voxel_generator.py:275-278 keeps the first max_points points of a voxel, in arrival order.  Until round 4 the (cell, point) pairs went
// through rocPRIM's stable radix sort: ~20 launches of its merge passes per call, memset nodes inside (the encoder had to stay in front
// of every hipGraph) and 2.7 % of the loop's kernel time.  No sort is needed:
//   cell_scan_*   exclusive prefix of the per-cell point counts (assign_kernel has them) -> every cell owns a segment of `seg`
//   seg_fill      every point drops its index into its cell's segment at the position assign_kernel's returning add gave it (arbitrary
//                 order inside the segment, no atomic)
//   voxel_slots   one wave per kept pillar: the max_points smallest indices of its segment in ascending order = its slots (arrival
//                 order).  A property of the index set: the result does not depend on the order the atomics ran in.
constexpr int kScanTile = 1024;
__global__ __launch_bounds__(kScanTile) void cell_scan_block_kernel(const int* __restrict__ count, size_t cells, int* __restrict__ seg_off,
                                                                    int* __restrict__ block_tot) {
    __shared__ int wsum[kScanTile / 64];
    const size_t i = (size_t)blockIdx.x * kScanTile + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = i < cells ? count[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (i < cells) seg_off[i] = base + incl - v;
    if (threadIdx.x == kScanTile - 1) block_tot[blockIdx.x] = base + incl;
}

// one block: exclusive prefix of the block totals (<= 4096 blocks = 4 M cells)
__global__ __launch_bounds__(1024) void cell_scan_tot_kernel(int* __restrict__ block_tot, int nblk) {
    // exclusive scan of <= 4096 block totals: 4 consecutive values per thread, wave scan, scan of the 16 wave totals (one thread walking
    // all totals took 12 us at 1024 of them)
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int v[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = t * 4 + k;
        v[k] = q < nblk ? block_tot[q] : 0;
        tot += v[k];
    }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    int run = base + incl - tot;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = t * 4 + k;
        if (q < nblk) block_tot[q] = run;
        run += v[k];
    }
}

__global__ __launch_bounds__(256) void seg_fill_kernel(const int* __restrict__ cell_of_point, int n_total, const int* __restrict__ seg_off,
                                                       const int* __restrict__ block_tot, const int* __restrict__ pos_in_cell,
                                                       int* __restrict__ seg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const int c = cell_of_point[i];
    if (c < 0) return;
    seg[seg_off[c] + block_tot[c / kScanTile] + pos_in_cell[i]] = i;  // (positions come from assign_kernel's returning adds: no atomic here)
}

// One wave per kept pillar: the max_points smallest indices of its segment, in ascending order = its slots.  n <= 64: every lane holds
// one index and counts the smaller ones (its arrival rank).  Crowded pillars: max_points rounds of "smallest index above the last one"
// over the whole segment (coalesced re-reads of a cached segment; 1 000 points: 16 loads per lane and round).  (A thread per POINT that
// walks its cell's segment -- the first form of this kernel -- took 144-165 us on the bench's clouds: the few pillars next to the sensor
// hold hundreds of points and every lane of their waves walks them all.)
constexpr int kSlotRegs = 16;  // voxel_slots_kernel: pillars of up to 1024 points are ranked from registers
__device__ __forceinline__ void voxel_slots_one(const int* __restrict__ coors, const int* __restrict__ num_voxels, int max_voxels, int gx, int gy,
                                                const int* __restrict__ seg_off, const int* __restrict__ block_tot,
                                                const int* __restrict__ count, const int* __restrict__ seg, int max_points,
                                                int* __restrict__ slots, long v, int lane) {
    const int b = (int)(v / max_voxels), ord = (int)(v % max_voxels);
    if (ord >= num_voxels[b]) return;
    const int c = (b * gx + coors[v * 4 + 2]) * gy + coors[v * 4 + 3];
    const int off = seg_off[c] + block_tot[c / kScanTile], n = count[c];
    int* out = slots + (size_t)v * max_points;
    if (n <= 64) {
        const int e = lane < n ? seg[off + lane] : 0x7fffffff;
        int r = 0;
        for (int k = 0; k < n; k++) r += __shfl(e, k) < e ? 1 : 0;
        if (lane < n && r < max_points) out[r] = e;
        return;
    }
    int last = -1;
    const int rounds = min(max_points, n);  // (max_points > n: the slots behind the n-th stay untouched, as on the n <= 64 path)
    if (n <= 64 * kSlotRegs) {
        // a crowded pillar (a wall stacks hundreds of returns into one pillar): the point indices are fetched ONCE into registers, the
        // `max_points` selection rounds run on them (the loop below re-reads the whole segment per round through a chain of dependent
        // global loads: ~0.5 us per round and 64 points -- one such wave set the launch's duration, 33 us)
        int x[kSlotRegs];
#pragma unroll
        for (int u = 0; u < kSlotRegs; u++) x[u] = lane + 64 * u < n ? seg[off + lane + 64 * u] : 0x7fffffff;
        for (int r = 0; r < rounds; r++) {
            int m = 0x7fffffff;
#pragma unroll
            for (int u = 0; u < kSlotRegs; u++) m = (x[u] > last && x[u] < m) ? x[u] : m;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
            if (lane == 0) out[r] = m;
            last = m;
        }
        return;
    }
    for (int r = 0; r < rounds; r++) {
        int m = 0x7fffffff;
        for (int j = lane; j < n; j += 64) {
            const int x = seg[off + j];
            m = (x > last && x < m) ? x : m;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
        if (lane == 0) out[r] = m;
        last = m;
    }
}

// a wave per pillar slot (most of the [batch, max_voxels] slots hold no pillar and exit at once; a capped grid whose waves walk the
// slots was measured slower: 35.7 vs 29.6 us at B = 4)
__global__ __launch_bounds__(256) void voxel_slots_kernel(const int* __restrict__ coors, const int* __restrict__ num_voxels, int max_voxels,
                                                          int batch, int gx, int gy, const int* __restrict__ seg_off,
                                                          const int* __restrict__ block_tot, const int* __restrict__ count,
                                                          const int* __restrict__ seg, int max_points, int* __restrict__ slots) {
    const int lane = threadIdx.x & 63;
    for (long v = (long)blockIdx.x * 4 + (threadIdx.x >> 6); v < (long)batch * max_voxels; v += (long)gridDim.x * 4)
        voxel_slots_one(coors, num_voxels, max_voxels, gx, gy, seg_off, block_tot, count, seg, max_points, slots, v, lane);
}

// ---------------------------------------------------------------------------------------------------------
// PFN.  The voxeliser leaves, per pillar, up to 20 point indices in first-come order (`slots`): reading a pillar is a
// chain of three dependent loads (count -> indices -> points).  `pfn_decorate` walks that chain ONCE, massively
// parallel (32 lanes per pillar, every pillar of the batch in flight), and writes the decorated feature rows
// [f_0..f_{F-1}, 1, pad] (12 floats) of all kept points contiguously in pillar order (CSR: pt_off[v] .. pt_off[v+1]).
// The three consumers -- BN statistics, forward, backward -- then stream rows: a wave stages the rows of kGroup
// consecutive pillars into LDS with coalesced 16-B loads and computes with lanes = 64 output channels.
// ---------------------------------------------------------------------------------------------------------
struct PfnGeom {
    float vx, vy, x_off, y_off, z_off;  // pillar_encoder.py:86-91 (offsets = v/2 + range_min)
};

__device__ __forceinline__ PfnGeom pfn_geom(const liso_pillar_cfg& c) {
    PfnGeom g;
    g.vx = c.vx; g.vy = c.vy;
    g.x_off = c.vx / 2 + c.x_min;
    g.y_off = c.vy / 2 + c.y_min;
    g.z_off = c.vz / 2 + c.z_min;
    return g;
}

constexpr int kGroup = 4;                       // pillars staged per wave step
constexpr int kStage = kGroup * kMaxPts;        // rows of LDS per wave (worst case)

// exclusive scan of the kept-point counts of all pillar rows (rows of samples beyond num_voxels count 0)
__device__ __forceinline__ int kept_points(const int* __restrict__ num_points, const int* __restrict__ num_voxels, int max_voxels,
                                           int rows, int v) {
    if (v >= rows) return 0;
    const int b = v / max_voxels;
    return (v - b * max_voxels) < num_voxels[b] ? num_points[v] : 0;
}

__global__ __launch_bounds__(1024) void pfn_scan_block_kernel(const int* __restrict__ num_points, const int* __restrict__ num_voxels,
                                                              int max_voxels, int rows, int* __restrict__ pt_off,
                                                              int* __restrict__ block_tot) {
    __shared__ int wsum[16];
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = kept_points(num_points, num_voxels, max_voxels, rows, i);
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if (i < rows) pt_off[i] = base + incl - v;
    if (threadIdx.x == 1023) block_tot[blockIdx.x] = base + incl;
}

__global__ __launch_bounds__(1024) void pfn_scan_tot_kernel(int* __restrict__ block_tot, int nblocks) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = (int)threadIdx.x < nblocks ? block_tot[threadIdx.x] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wsum[w];
    if ((int)threadIdx.x < nblocks) block_tot[threadIdx.x] = base + incl - v;
    if ((int)threadIdx.x == nblocks - 1) block_tot[nblocks] = base + incl;
}

__global__ __launch_bounds__(1024) void pfn_scan_add_kernel(int* __restrict__ pt_off, int rows, const int* __restrict__ block_tot,
                                                            int nblocks) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < rows) pt_off[i] += block_tot[blockIdx.x];
    if (i == 0) pt_off[rows] = block_tot[nblocks];
}

// Decorated rows.  Channel order (pillar_encoder.py:109-146 with legacy aliasing, :129-139): the in-place f_center update
// overwrites xyz of `features` itself, so the 10 inputs are [fc(3), extras(C-3), cluster(3), fc(3)], and because
// of the x/y swap at pcl_to_feature_grid.py:73, fc_x uses the *y index* and fc_y the *x index*.
template <int C>
__global__ __launch_bounds__(256) void pfn_decorate_kernel(const float* __restrict__ pts, liso_pillar_cfg cfg, int rows,
                                                           const int* __restrict__ coors, const int* __restrict__ num_points,
                                                           const int* __restrict__ slots, const int* __restrict__ num_voxels,
                                                           const int* __restrict__ pt_off, float* __restrict__ feat,
                                                           int* __restrict__ voxel_cell) {
#pragma clang fp contract(off)
    constexpr int F = C + 6;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int v = (int)(t >> 5), s = (int)(t & 31);  // 32 lanes per pillar (max_points <= 32)
    if (v >= rows) return;
    const int num = kept_points(num_points, num_voxels, cfg.max_voxels, rows, v);
    const int xi = coors[v * 4 + 2], yi = coors[v * 4 + 3];
    if (s == 0) voxel_cell[v] = num > 0 ? ((v / cfg.max_voxels) * cfg.gx + xi) * cfg.gy + yi : -1;
    float p[C];
#pragma unroll
    for (int k = 0; k < C; k++) p[k] = 0.f;
    if (s < num) {
        const int idx = slots[(size_t)v * cfg.max_points + s];
        const float* q = pts + (size_t)idx * C;
        if constexpr (C == 4) {
            const float4 tt = *reinterpret_cast<const float4*>(q);
            p[0] = tt.x; p[1] = tt.y; p[2] = tt.z; p[3] = tt.w;
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) p[k] = q[k];
        }
    }
    // points_mean = sum over the (zero padded) slots / num_points, pillar_encoder.py:112-115; the butterfly offsets
    // stay inside the 32-lane half that holds this pillar
    float sx = p[0], sy = p[1], sz = p[2];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        sz += __shfl_xor(sz, o);
    }
    if (s >= num) return;
    const PfnGeom g = pfn_geom(cfg);
    const float fn = (float)num;
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;
    const float fcx = p[0] - ((float)yi * g.vx + g.x_off);  // coors[:,3] (= y index) * vx, :130-132
    const float fcy = p[1] - ((float)xi * g.vy + g.y_off);  // coors[:,2] (= x index) * vy, :133-135
    const float fcz = p[2] - (0.f + g.z_off);               // coors[:,1] == 0,              :136-138
    float f[kFP];
#pragma unroll
    for (int k = 0; k < kFP; k++) f[k] = 0.f;
    f[0] = fcx; f[1] = fcy; f[2] = fcz;
#pragma unroll
    for (int k = 3; k < C; k++) f[k] = p[k];
    f[C + 0] = p[0] - mx; f[C + 1] = p[1] - my; f[C + 2] = p[2] - mz;
    f[C + 3] = fcx; f[C + 4] = fcy; f[C + 5] = fcz;
    f[F] = 1.f;
    float4* row = reinterpret_cast<float4*>(feat + ((size_t)pt_off[v] + s) * kFP);
    row[0] = make_float4(f[0], f[1], f[2], f[3]);
    row[1] = make_float4(f[4], f[5], f[6], f[7]);
    row[2] = make_float4(f[8], f[9], f[10], f[11]);
}

// ---- stats: second moments of the augmented feature vector [f, 1] over all kept rows ------------------------
// Rows are independent: every block takes a contiguous chunk, a wave stages 64 rows at a time, lanes = moment pairs.
template <int C>
__global__ __launch_bounds__(kPfnThreads) void pfn_stats_kernel(const float* __restrict__ feat, const int* __restrict__ pt_off,
                                                                int rows, double* __restrict__ partials) {
    constexpr int F = C + 6, D = F + 1, NP = D * (D + 1) / 2;
    __shared__ __attribute__((aligned(16))) float tile[kPfnThreads / 64][64][kFP];
    __shared__ double red[kPfnThreads / 64][LISO_PFN_STATS_DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // lane -> up to two (j,k) pairs of the upper triangle
    int pj[2] = {0, 0}, pk[2] = {0, 0};
    {
        int idx = 0;
        for (int j = 0; j < D; j++)
            for (int k = j; k < D; k++, idx++) {
                if (idx == lane) { pj[0] = j; pk[0] = k; }
                if (idx == lane + 64) { pj[1] = j; pk[1] = k; }
            }
    }
    const long n = pt_off[rows];
    const long waves = (long)gridDim.x * (kPfnThreads / 64);
    const long per = ((n + waves - 1) / waves + 63) / 64 * 64;  // rows per wave, multiple of the 64-row tile
    const long w_id = (long)blockIdx.x * (kPfnThreads / 64) + wave;
    const long r0 = w_id * per, r1 = r0 + per < n ? r0 + per : n;
    double acc0 = 0.0, acc1 = 0.0;
    for (long base = r0; base < r1; base += 64) {
        const int cnt = (int)(r1 - base < 64 ? r1 - base : 64);
        const float4* src = reinterpret_cast<const float4*>(feat + (size_t)base * kFP);
        float4* dst = reinterpret_cast<float4*>(&tile[wave][0][0]);
        for (int i = lane; i < cnt * 3; i += 64) dst[i] = src[i];
        __builtin_amdgcn_wave_barrier();
        for (int s = 0; s < cnt; s++) {
            const float* f = tile[wave][s];
            acc0 += (double)f[pj[0]] * (double)f[pk[0]];
            if (lane + 64 < NP) acc1 += (double)f[pj[1]] * (double)f[pk[1]];
        }
        __builtin_amdgcn_wave_barrier();
    }
    red[wave][lane] = acc0;
    if (lane + 64 < LISO_PFN_STATS_DOUBLES) red[wave][lane + 64] = lane + 64 < NP ? acc1 : 0.0;
    __syncthreads();
    if (threadIdx.x < LISO_PFN_STATS_DOUBLES) {
        double s = 0.0;
        for (int w = 0; w < kPfnThreads / 64; w++) s += red[w][threadIdx.x];
        partials[(size_t)blockIdx.x * LISO_PFN_STATS_DOUBLES + threadIdx.x] = threadIdx.x < NP ? s : 0.0;
    }
}

__device__ __forceinline__ int pair_index(int j, int k, int D) {  // j <= k, row-major upper triangle
    return j * D - j * (j - 1) / 2 + (k - j);
}

// reduce partials (fixed order), then per-channel mean/var -> scale/shift (+ running stats), utils.py:166-168
template <int C>
__global__ __launch_bounds__(1024) void pfn_bn_finalize_kernel(const double* __restrict__ partials, int nblocks,
                                                               const int* __restrict__ num_voxels, int batch,
                                                               int max_points, const float* __restrict__ weight,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float momentum,
                                                               float eps, float* __restrict__ bn_out,
                                                               double* __restrict__ moments) {
    constexpr int F = C + 6, D = F + 1;
    constexpr int NS = LISO_PFN_STATS_DOUBLES, CH = 1024 / NS;  // 13 chunks of blocks x 78 sums (8 x 128 left 50 lanes of every chunk idle)
    __shared__ double part[CH][NS];
    __shared__ double mom[NS];
    const int e = threadIdx.x % NS, chunk = threadIdx.x / NS;
    if (chunk < CH) {
        double s = 0.0;
        const int per = (nblocks + CH - 1) / CH;
        const int lo = chunk * per, hi = lo + per < nblocks ? lo + per : nblocks;
        int blk = lo;
        for (; blk + 16 <= hi; blk += 16) {  // 16 independent loads in flight, summed in block order
            double v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = partials[(size_t)(blk + j) * NS + e];
#pragma unroll
            for (int j = 0; j < 16; j++) s += v[j];
        }
        for (; blk < hi; blk++) s += partials[(size_t)blk * NS + e];
        part[chunk][e] = s;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0.0;
        for (int c = 0; c < CH; c++) t += part[c][threadIdx.x];
        mom[threadIdx.x] = t;
        moments[threadIdx.x] = t;
    }
    __syncthreads();
    // the quadratic form w^T M w of every output channel: thread (channel, j) takes row j (F products), the F row sums of a channel are
    // added in row order (one thread per channel walking all F x F products through LDS took ~10 us of this launch)
    __shared__ double rows_sq[kOut][F + 1], rows_sum[kOut][F + 1];
    if (threadIdx.x < kOut * F) {
        const int c = threadIdx.x / F, j = threadIdx.x % F;
        const double wj = (double)weight[c * F + j];
        double q = 0.0;
        for (int k = 0; k < F; k++) {
            const int a = j <= k ? j : k, bb = j <= k ? k : j;
            q += wj * (double)weight[c * F + k] * mom[pair_index(a, bb, D)];
        }
        rows_sq[c][j] = q;
        rows_sum[c][j] = wj * mom[pair_index(j, D - 1, D)];
    }
    __syncthreads();
    if (threadIdx.x < kOut) {
        const int c = threadIdx.x;
        long long P = 0;
        for (int b = 0; b < batch; b++) P += num_voxels[b];
        const double M = (double)P * (double)max_points;  // BatchNorm1d sees [P, 64, 20]: padded rows count
        double sum = 0.0, sq = 0.0;
        for (int j = 0; j < F; j++) {
            sum += rows_sum[c][j];
            sq += rows_sq[c][j];
        }
        double mean = 0.0, var = 0.0;
        if (M > 0.0) {
            mean = sum / M;
            var = sq / M - mean * mean;
            if (var < 0.0) var = 0.0;
        }
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const float scale = (float)((double)gamma[c] * invstd);
        bn_out[c] = scale;
        bn_out[kOut + c] = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
        bn_out[2 * kOut + c] = (float)mean;
        bn_out[3 * kOut + c] = (float)invstd;
        if (M > 1.0) {  // torch.nn.BatchNorm1d running stats: unbiased variance
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * M / (M - 1.0));
        }
    }
}

__global__ void pfn_bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                   float eps, float* __restrict__ bn_out) {
    const int c = threadIdx.x;
    if (c >= kOut) return;
    const float invstd = 1.f / sqrtf(running_var[c] + eps);
    bn_out[c] = gamma[c] * invstd;
    bn_out[kOut + c] = beta[c] - running_mean[c] * gamma[c] * invstd;
    bn_out[2 * kOut + c] = running_mean[c];
    bn_out[3 * kOut + c] = invstd;
}

template <typename T> __device__ __forceinline__ void store_out(T* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }
template <typename T> __device__ __forceinline__ float load_in(const T* p);
template <> __device__ __forceinline__ float load_in<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_in<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }

// stage the rows of pillars [v0, v0 + kGroup) of one wave step: returns the group's first row offset, fills off[] (lane
// g holds pt_off[v0+g] for g <= kGroup) and copies the rows into `frow` with 16-B loads
__device__ __forceinline__ void stage_rows(const float* __restrict__ feat, const int* __restrict__ pt_off, int rows, int v0, int lane,
                                           float (*frow)[kFP], int (&off)[kGroup + 1]) {
    int o = 0;
    if (lane <= kGroup) { const int v = v0 + lane; o = pt_off[v < rows ? v : rows]; }
#pragma unroll
    for (int g = 0; g <= kGroup; g++) off[g] = __shfl(o, g);
    const int total = off[kGroup] - off[0];
    const float4* src = reinterpret_cast<const float4*>(feat + (size_t)off[0] * kFP);
    float4* dst = reinterpret_cast<float4*>(&frow[0][0]);
    for (int i = lane; i < total * 3; i += 64) dst[i] = src[i];
}

// ---- forward: Linear(F,64, no bias) -> BN -> ReLU -> max over the 20 slots -> canvas[b, x_idx, y_idx, :] ------------
// One launch, two kinds of blocks, every canvas cell written exactly once (no memset pass, no double write):
//   the first blocks:        pillars in voxel order, kGroup per wave step, rows streamed from the CSR feature array,
//                            one 128-B (bf16) / 256-B (fp32) row store per pillar (dependent loads, little bandwidth: ~25 us
//                            of work at B = 4 that is scheduled first so it runs underneath the streaming blocks);
//   the last zero_blocks:    256 consecutive cells each; empty cells are zero-filled with 16-B stores (consecutive
//                            threads -> consecutive chunks), the occupancy map is written (alone: 21 us = 6.4 TB/s at B = 4).
constexpr int kCellsPerBlock = 256;

template <int C, typename OutT>
__global__ __launch_bounds__(kPfnThreads) void pfn_forward_kernel(const float* __restrict__ feat, const int* __restrict__ pt_off,
                                                                  const int* __restrict__ voxel_cell, int rows, int max_points,
                                                                  long total_cells, int zero_blocks,
                                                                  const int* __restrict__ cell_to_voxel,
                                                                  const float* __restrict__ weight, const float* __restrict__ bn,
                                                                  OutT* __restrict__ canvas, float* __restrict__ occupancy) {
    constexpr int F = C + 6;
    constexpr int kChunks = kOut * (int)sizeof(OutT) / 16;  // 16-B chunks per cell (8 for bf16, 16 for fp32)
    __shared__ __attribute__((aligned(16))) float frow[kPfnThreads / 64][kStage][kFP];
    __shared__ int vox[kCellsPerBlock];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pillar_blocks = (int)gridDim.x - zero_blocks;
    if ((int)blockIdx.x >= pillar_blocks) {
        const long cell0 = (long)((int)blockIdx.x - pillar_blocks) * kCellsPerBlock;
        const long cell = cell0 + tid;
        int v1 = 0;
        if (cell < total_cells) {
            v1 = cell_to_voxel[cell];
            occupancy[cell] = v1 > 0 ? 1.f : 0.f;
        }
        vox[tid] = v1;
        __syncthreads();
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        uint4* base = reinterpret_cast<uint4*>(canvas + (size_t)cell0 * kOut);
        for (int id = tid; id < kCellsPerBlock * kChunks; id += kPfnThreads) {
            const int cl = id / kChunks;
            if (cell0 + cl < total_cells && vox[cl] == 0) base[id] = z;
        }
        return;
    }
    float w[F];
#pragma unroll
    for (int k = 0; k < F; k++) w[k] = weight[lane * F + k];
    const float scale = bn[lane], shift = bn[kOut + lane];
    const float pad_val = fmaxf(shift, 0.f);
    const int wave_id = (int)blockIdx.x * (kPfnThreads / 64) + wave;
    const int n_waves = pillar_blocks * (kPfnThreads / 64);
    for (int v0 = wave_id * kGroup; v0 < rows; v0 += n_waves * kGroup) {
        int off[kGroup + 1];
        stage_rows(feat, pt_off, rows, v0, lane, frow[wave], off);
        int cell_l = -1;
        if (lane < kGroup && v0 + lane < rows) cell_l = voxel_cell[v0 + lane];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < kGroup; g++) {
            const int num = off[g + 1] - off[g];
            const int cell = __shfl(cell_l, g);
            if (num <= 0 || cell < 0) continue;
            float best = num < max_points ? pad_val : 0.f;
            const int r0 = off[g] - off[0];
            for (int s = 0; s < num; s++) {
                const float4* f4 = reinterpret_cast<const float4*>(frow[wave][r0 + s]);
                const float4 a = f4[0], b = f4[1], c4 = f4[2];
                const float f[kFP] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c4.x, c4.y, c4.z, c4.w};
                float x = 0.f;
#pragma unroll
                for (int k = 0; k < F; k++) x = fmaf(w[k], f[k], x);
                best = fmaxf(best, fmaxf(fmaf(x, scale, shift), 0.f));
            }
            store_out<OutT>(canvas + (size_t)cell * kOut + lane, best);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- backward: per-block partial sums of  A[c][k] = sum dz*f,  dbeta[c] = sum dz,  dgamma[c] = sum dz*xhat ------
template <int C, typename GT>
__global__ __launch_bounds__(kPfnThreads) void pfn_backward_kernel(const float* __restrict__ feat, const int* __restrict__ pt_off,
                                                                   const int* __restrict__ voxel_cell, int rows, int max_points,
                                                                   const float* __restrict__ weight, const float* __restrict__ bn,
                                                                   const GT* __restrict__ grad_canvas,
                                                                   float* __restrict__ partials) {
    constexpr int F = C + 6, NA = F + 2;
    __shared__ __attribute__((aligned(16))) float frow[kPfnThreads / 64][kStage][kFP];
    __shared__ float red[kPfnThreads / 64][NA][kOut];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[F], acc[NA];
#pragma unroll
    for (int k = 0; k < F; k++) w[k] = weight[lane * F + k];
#pragma unroll
    for (int k = 0; k < NA; k++) acc[k] = 0.f;
    const float scale = bn[lane], shift = bn[kOut + lane], mean = bn[2 * kOut + lane], invstd = bn[3 * kOut + lane];
    const float pad_val = fmaxf(shift, 0.f);
    const int wave_id = (int)blockIdx.x * (kPfnThreads / 64) + wave;
    const int n_waves = (int)gridDim.x * (kPfnThreads / 64);
    for (int v0 = wave_id * kGroup; v0 < rows; v0 += n_waves * kGroup) {
        int off[kGroup + 1];
        stage_rows(feat, pt_off, rows, v0, lane, frow[wave], off);
        int cell_l = -1;
        if (lane < kGroup && v0 + lane < rows) cell_l = voxel_cell[v0 + lane];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < kGroup; g++) {
            const int num = off[g + 1] - off[g];
            const int cell = __shfl(cell_l, g);
            if (num <= 0 || cell < 0) continue;
            const int r0 = off[g] - off[0];
            float best = -1.f, best_x = 0.f;
            int best_s = -1;
            for (int s = 0; s < num; s++) {
                const float* f = frow[wave][r0 + s];
                float x = 0.f;
#pragma unroll
                for (int k = 0; k < F; k++) x = fmaf(w[k], f[k], x);
                const float y = fmaxf(fmaf(x, scale, shift), 0.f);
                if (y > best) { best = y; best_s = s; best_x = x; }
            }
            if (num < max_points && pad_val > best) { best = pad_val; best_s = -1; best_x = 0.f; }
            const float gy = load_in<GT>(grad_canvas + (size_t)cell * kOut + lane);
            const float dz = best > 0.f ? gy : 0.f;  // ReLU gate; the max routes the gradient to one row
            if (best_s >= 0) {
                const float* f = frow[wave][r0 + best_s];
#pragma unroll
                for (int k = 0; k < F; k++) acc[k] = fmaf(dz, f[k], acc[k]);
            }
            acc[F] += dz;
            acc[F + 1] = fmaf(dz, (best_x - mean) * invstd, acc[F + 1]);
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int k = 0; k < NA; k++) red[wave][k][lane] = acc[k];
    __syncthreads();
    for (int idx = threadIdx.x; idx < NA * kOut; idx += kPfnThreads) {
        const int k = idx / kOut, c = idx % kOut;
        float s = 0.f;
        for (int wv = 0; wv < kPfnThreads / 64; wv++) s += red[wv][k][c];
        partials[((size_t)blockIdx.x * NA + k) * kOut + c] = s;
    }
}

// stage 1: one block per accumulator row k, 64 channels x 16 chunks of block partials, fixed combination order
template <int NA>
__global__ __launch_bounds__(1024) void pfn_backward_reduce_kernel(const float* __restrict__ partials, int nblocks,
                                                                   double* __restrict__ tot) {
    __shared__ double part[16][kOut];
    const int k = blockIdx.x, c = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    const int per = (nblocks + 15) / 16;
    const int lo = chunk * per, hi = lo + per < nblocks ? lo + per : nblocks;
    double s = 0.0;
    {
        int blk = lo;
        for (; blk + 8 <= hi; blk += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = partials[((size_t)(blk + j) * NA + k) * kOut + c];
#pragma unroll
            for (int j = 0; j < 8; j++) s += (double)v[j];
        }
        for (; blk < hi; blk++) s += (double)partials[((size_t)blk * NA + k) * kOut + c];
    }
    part[chunk][c] = s;
    __syncthreads();
    if (chunk == 0) {
        double t = 0.0;
        for (int q = 0; q < 16; q++) t += part[q][c];
        tot[k * kOut + c] = t;
    }
}

// stage 2: closed-form BN backward on the reduced sums (training) / plain scale (eval)
template <int C>
__global__ __launch_bounds__(64) void pfn_backward_finalize_kernel(const double* __restrict__ tot,
                                                                   const int* __restrict__ num_voxels, int batch,
                                                                   int max_points, const float* __restrict__ weight,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ bn,
                                                                   const double* __restrict__ moments, int training,
                                                                   float* __restrict__ grad_weight,
                                                                   float* __restrict__ grad_gamma,
                                                                   float* __restrict__ grad_beta) {
    constexpr int F = C + 6, D = F + 1;
    const int c = threadIdx.x;
    if (c >= kOut) return;
    const double dbeta = tot[F * kOut + c], dgamma = tot[(F + 1) * kOut + c];
    grad_beta[c] = (float)dbeta;
    grad_gamma[c] = (float)dgamma;
    const double invstd = (double)bn[3 * kOut + c], mean = (double)bn[2 * kOut + c];
    const double gi = (double)gamma[c] * invstd;
    if (!training) {
        for (int k = 0; k < F; k++) grad_weight[c * F + k] = (float)(gi * tot[k * kOut + c]);
        return;
    }
    long long P = 0;
    for (int b = 0; b < batch; b++) P += num_voxels[b];
    const double M = (double)P * (double)max_points;
    double w[F];
    for (int k = 0; k < F; k++) w[k] = (double)weight[c * F + k];
    for (int k = 0; k < F; k++) {
        const double Fk = moments[pair_index(k, D - 1, D)];  // sum_rows f_k
        double xf = 0.0;                                      // sum_rows x_c f_k = sum_j w_j FF[j][k]
        for (int j = 0; j < F; j++) {
            const int a = j <= k ? j : k, bb = j <= k ? k : j;
            xf += w[j] * moments[pair_index(a, bb, D)];
        }
        const double xhat_f = invstd * (xf - mean * Fk);      // sum_rows xhat_c f_k
        const double dw = M > 0.0 ? gi * (tot[k * kOut + c] - dbeta / M * Fk - dgamma / M * xhat_f) : 0.0;
        grad_weight[c * F + k] = (float)dw;
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

inline bool cfg_ok(const liso_pillar_cfg* c, int batch) {
    return c && batch >= 1 && batch <= LISO_PILLARS_MAX_BATCH && c->gx > 0 && c->gy > 0 && c->max_points >= 1 &&
           c->max_points <= kMaxPts && c->max_voxels >= 1 && c->n_channels >= 3 && c->n_channels <= 5;
}

inline BatchInfo make_batch(const int* off, int batch) {
    BatchInfo bi;
    bi.tile_off[0] = 0;
    for (int b = 0; b <= LISO_PILLARS_MAX_BATCH; b++) bi.off[b] = off[b <= batch ? b : batch];
    for (int b = 0; b < LISO_PILLARS_MAX_BATCH; b++) {
        const int n = b < batch ? off[b + 1] - off[b] : 0;
        bi.tile_off[b + 1] = bi.tile_off[b] + (n + kTile - 1) / kTile;
    }
    return bi;
}

inline int pfn_grid(int rows) {
    const int need = (rows + kPfnThreads / 64 - 1) / (kPfnThreads / 64);
    return need < kPfnGrid ? (need > 0 ? need : 1) : kPfnGrid;
}

}  // namespace

extern "C" {

size_t liso_pillars_voxelize_workspace_bytes(const liso_pillar_cfg* cfg, int batch, int n_total) {
    if (!cfg_ok(cfg, batch) || n_total < 0) return 0;
    const size_t cells = (size_t)batch * cfg->gx * cfg->gy;
    const size_t tiles = (size_t)(n_total + kTile - 1) / kTile + batch;
    // count | first_enc | seg_off [cells each], cell_of_point | pos_in_cell | seg [n each], tile counts, scan block totals
    if (cells > (size_t)4096 * kScanTile) return 0;  // (the one-block pass over the scan's block totals holds 4096 of them: 4 M cells)
    return (3 * cells + 3 * (size_t)n_total + tiles + 64 + 4096) * sizeof(int) + 256;
}

int liso_pillars_voxelize_f32(const float* points, const int* offsets_host, int batch, const liso_pillar_cfg* cfg,
                              int* coors, int* num_points, int* slots, int* num_voxels, int* cell_to_voxel,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (!cfg_ok(cfg, batch) || !offsets_host) return LISO_EINVAL;
    const int n_total = offsets_host[batch];
    if (n_total < 0 || offsets_host[0] != 0) return LISO_EINVAL;
    for (int b = 0; b < batch; b++)
        if (offsets_host[b + 1] < offsets_host[b]) return LISO_EINVAL;
    if (!coors || !num_points || !slots || !num_voxels || !cell_to_voxel || !workspace || (n_total > 0 && !points))
        return LISO_EINVAL;
    if ((size_t)batch * cfg->gx * cfg->gy > (size_t)4096 * kScanTile) return LISO_EINVAL;  // (see liso_pillars_voxelize_workspace_bytes)
    if (workspace_bytes < liso_pillars_voxelize_workspace_bytes(cfg, batch, n_total)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)batch * cfg->gx * cfg->gy;
    int* count = (int*)workspace;
    int* first_enc = count + cells;
    int* cell_of_point = first_enc + cells;
    int* tile_count = cell_of_point + n_total;
    const size_t tiles_cap = (size_t)(n_total + kTile - 1) / kTile + batch;
    int* seg_off = tile_count + tiles_cap + 32;
    int* pos_in_cell = seg_off + cells;
    int* seg = pos_in_cell + n_total;
    int* block_tot = seg + n_total;
    const BatchInfo bi = make_batch(offsets_host, batch);
    const int tiles = bi.tile_off[batch];
    if (liso_zero::zero_async(count, 2 * cells * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (liso_zero::zero_async(cell_to_voxel, cells * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (liso_zero::zero_async(num_voxels, batch * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (n_total == 0) return LISO_OK;
    const int nb = (n_total + 255) / 256;
    hipLaunchKernelGGL(assign_kernel, dim3(nb), dim3(256), 0, st, points, n_total, cfg->n_channels, bi, batch, *cfg,
                       cell_of_point, count, first_enc, pos_in_cell);
    hipLaunchKernelGGL(tile_count_kernel, dim3(tiles), dim3(kTile), 0, st, cell_of_point, first_enc, bi, batch,
                       tile_count);
    hipLaunchKernelGGL(rank_kernel, dim3(tiles), dim3(kTile), 0, st, cell_of_point, first_enc, count, bi, batch, *cfg,
                       tile_count, coors, num_points, cell_to_voxel, num_voxels);
    const int scan_blocks = (int)((cells + kScanTile - 1) / kScanTile);
    cell_scan_block_kernel<<<scan_blocks, kScanTile, 0, st>>>(count, cells, seg_off, block_tot);
    cell_scan_tot_kernel<<<1, 1024, 0, st>>>(block_tot, scan_blocks);
    seg_fill_kernel<<<nb, 256, 0, st>>>(cell_of_point, n_total, seg_off, block_tot, pos_in_cell, seg);
    const long n_wave = (long)batch * cfg->max_voxels;
    voxel_slots_kernel<<<(unsigned)((n_wave + 3) / 4), 256, 0, st>>>(coors, num_voxels, cfg->max_voxels, batch, cfg->gx, cfg->gy, seg_off,
                                                                   block_tot, count, seg, cfg->max_points, slots);
    return check_launch();
}

size_t liso_pfn_partials_bytes(void) {
    const size_t stats = (size_t)kPfnGrid * LISO_PFN_STATS_DOUBLES * sizeof(double);
    const size_t bwd = (size_t)kPfnGrid * (11 + 2) * kOut * sizeof(float) + (11 + 2) * kOut * sizeof(double);
    return stats > bwd ? stats : bwd;
}

size_t liso_pfn_decorate_workspace_bytes(int batch, int max_voxels) {
    if (batch < 1 || max_voxels < 1) return 0;
    return ((size_t)(batch * (long)max_voxels + 1023) / 1024 + 2) * sizeof(int);
}

int liso_pfn_decorate_f32(const float* points, const liso_pillar_cfg* cfg, int batch, const int* coors, const int* num_points,
                          const int* slots, const int* num_voxels, int* pt_off, float* feat, int* voxel_cell, void* workspace,
                          size_t workspace_bytes, void* stream) {
    // `points` is only dereferenced for kept points: it may be NULL when the batch holds no point at all
    if (!cfg_ok(cfg, batch) || !coors || !num_points || !slots || !num_voxels || !pt_off || !feat || !voxel_cell || !workspace)
        return LISO_EINVAL;
    if (workspace_bytes < liso_pfn_decorate_workspace_bytes(batch, cfg->max_voxels)) return LISO_EWORKSPACE;
    const int rows = batch * cfg->max_voxels;
    const int nsb = (rows + 1023) / 1024;
    if (nsb > 1024) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int* block_tot = (int*)workspace;
    pfn_scan_block_kernel<<<nsb, 1024, 0, st>>>(num_points, num_voxels, cfg->max_voxels, rows, pt_off, block_tot);
    pfn_scan_tot_kernel<<<1, 1024, 0, st>>>(block_tot, nsb);
    pfn_scan_add_kernel<<<nsb, 1024, 0, st>>>(pt_off, rows, block_tot, nsb);
    const unsigned grid = (unsigned)(((long)rows * 32 + 255) / 256);
    switch (cfg->n_channels) {
        case 3: pfn_decorate_kernel<3><<<grid, 256, 0, st>>>(points, *cfg, rows, coors, num_points, slots, num_voxels, pt_off, feat, voxel_cell); break;
        case 4: pfn_decorate_kernel<4><<<grid, 256, 0, st>>>(points, *cfg, rows, coors, num_points, slots, num_voxels, pt_off, feat, voxel_cell); break;
        default: pfn_decorate_kernel<5><<<grid, 256, 0, st>>>(points, *cfg, rows, coors, num_points, slots, num_voxels, pt_off, feat, voxel_cell); break;
    }
    return check_launch();
}

int liso_pfn_bn_prepare_f32(const float* feat, const int* pt_off, const liso_pillar_cfg* cfg, int batch, const int* num_voxels,
                            const float* weight, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, int training, float* bn_out, double* moments,
                            void* partials, void* stream) {
    if (!cfg_ok(cfg, batch) || !gamma || !beta || !running_mean || !running_var || !bn_out) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!training) {
        pfn_bn_eval_kernel<<<1, 64, 0, st>>>(gamma, beta, running_mean, running_var, eps, bn_out);
        return check_launch();
    }
    if (!feat || !pt_off || !num_voxels || !weight || !moments || !partials) return LISO_EINVAL;
    const int rows = batch * cfg->max_voxels;
    const int grid = kPfnGrid;
#define LISO_PREP(CC)                                                                                                     \
    do {                                                                                                                  \
        pfn_stats_kernel<CC><<<grid, kPfnThreads, 0, st>>>(feat, pt_off, rows, (double*)partials);                           \
        pfn_bn_finalize_kernel<CC><<<1, 1024, 0, st>>>((const double*)partials, grid, num_voxels, batch, cfg->max_points,    \
                                                      weight, gamma, beta, running_mean, running_var, momentum, eps, bn_out, \
                                                      moments);                                                           \
    } while (0)
    switch (cfg->n_channels) {
        case 3: LISO_PREP(3); break;
        case 4: LISO_PREP(4); break;
        default: LISO_PREP(5); break;
    }
#undef LISO_PREP
    return check_launch();
}

int liso_pfn_forward_scatter(const float* feat, const int* pt_off, const int* voxel_cell, const liso_pillar_cfg* cfg, int batch,
                             const int* cell_to_voxel, const float* weight, const float* bn_out, void* canvas, int out_bf16,
                             float* occupancy, void* stream) {
    if (!cfg_ok(cfg, batch) || !feat || !pt_off || !voxel_cell || !cell_to_voxel || !weight || !bn_out || !canvas || !occupancy)
        return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long cells = (long)batch * cfg->gx * cfg->gy;
    const int rows = batch * cfg->max_voxels;
    const int zero_blocks = (int)((cells + kCellsPerBlock - 1) / kCellsPerBlock);
    const int groups = (rows + kGroup - 1) / kGroup;
    int pfn_blocks = (groups + kPfnThreads / 64 - 1) / (kPfnThreads / 64);
    if (pfn_blocks > 4096) pfn_blocks = 4096;
    const unsigned grid = (unsigned)(zero_blocks + pfn_blocks);
#define LISO_FWD(CC, T) pfn_forward_kernel<CC, T><<<grid, kPfnThreads, 0, st>>>(feat, pt_off, voxel_cell, rows, cfg->max_points, cells, zero_blocks, cell_to_voxel, weight, bn_out, (T*)canvas, occupancy)
    switch (cfg->n_channels * 2 + (out_bf16 ? 1 : 0)) {
        case 6: LISO_FWD(3, float); break;
        case 7: LISO_FWD(3, __hip_bfloat16); break;
        case 8: LISO_FWD(4, float); break;
        case 9: LISO_FWD(4, __hip_bfloat16); break;
        case 10: LISO_FWD(5, float); break;
        default: LISO_FWD(5, __hip_bfloat16); break;
    }
#undef LISO_FWD
    return check_launch();
}

int liso_pfn_backward(const float* feat, const int* pt_off, const int* voxel_cell, const liso_pillar_cfg* cfg, int batch,
                      const int* num_voxels, const float* weight, const float* gamma, const float* bn_out,
                      const double* moments, int training, const void* grad_canvas, int grad_bf16, float* grad_weight,
                      float* grad_gamma, float* grad_beta, void* partials, void* stream) {
    if (!cfg_ok(cfg, batch) || !feat || !pt_off || !voxel_cell || !num_voxels || !weight || !gamma || !bn_out || !grad_canvas ||
        !grad_weight || !grad_gamma || !grad_beta || !partials || (training && !moments))
        return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rows = batch * cfg->max_voxels;
    const int grid = pfn_grid((rows + kGroup - 1) / kGroup);
#define LISO_BWD(CC, T)                                                                                              \
    do {                                                                                                             \
        constexpr int NA_ = CC + 8;                                                                                  \
        double* tot_ = (double*)((float*)partials + (size_t)kPfnGrid * (11 + 2) * kOut);                             \
        pfn_backward_kernel<CC, T><<<grid, kPfnThreads, 0, st>>>(feat, pt_off, voxel_cell, rows, cfg->max_points,     \
                                                                 weight, bn_out, (const T*)grad_canvas,             \
                                                                 (float*)partials);                                  \
        pfn_backward_reduce_kernel<NA_><<<NA_, 1024, 0, st>>>((const float*)partials, grid, tot_);                    \
        pfn_backward_finalize_kernel<CC><<<1, 64, 0, st>>>(tot_, num_voxels, batch, cfg->max_points, weight, gamma,   \
                                                           bn_out, moments, training, grad_weight, grad_gamma,       \
                                                           grad_beta);                                               \
    } while (0)
    switch (cfg->n_channels * 2 + (grad_bf16 ? 1 : 0)) {
        case 6: LISO_BWD(3, float); break;
        case 7: LISO_BWD(3, __hip_bfloat16); break;
        case 8: LISO_BWD(4, float); break;
        case 9: LISO_BWD(4, __hip_bfloat16); break;
        case 10: LISO_BWD(5, float); break;
        default: LISO_BWD(5, __hip_bfloat16); break;
    }
#undef LISO_BWD
    return check_launch();
}

}  // extern "C"
