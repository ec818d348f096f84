// Pillar path for gfx950 (MI355X): hard voxelisation -> fused PillarFeatureNet -> dense BEV scatter (+ backward).
// C ABI and the reference lines each stage replaces: include/liso_pillars.h.
//
// Design (HBM-bound stage; nothing here is GEMM-shaped enough for MFMA: K = C+6 <= 11):
//   * voxelise is deterministic and sort-free:
//       assign      per point: cell id, atomicAdd(count[cell]), atomicMax(first[cell], INT_MAX - i)
//       tile_count  per 1024-point tile: how many points are the first of their cell
//       rank        voxel ordinal = exclusive prefix of "is first" in point order (the reference's voxel order),
//                   voxels >= max_voxels dropped, coors / num_points / cell->row written
//       fill        each point inserts its index into its voxel's 20 slots with an atomicMin cascade; the final
//                   state is the 20 smallest indices in ascending order whatever the execution order
//   * the PFN never materialises voxels[P,20,C], [P,20,10] or [P,20,64]: one wavefront owns one pillar, lanes 0..19
//     gather the points (16 B each, L2-resident cloud), build the 10 decorated features into LDS, then the 64 lanes
//     ARE the 64 output channels: 10 FMAs per row against an LDS-broadcast feature row, running max in a register,
//     one coalesced 64-channel store per pillar into the channels-last canvas.
//   * BatchNorm1d batch statistics come from the second moments of the 10-vector (sum f f^T in fp64, 66 numbers)
//     instead of 2x64 per-channel sums over [P*20, 64]; the same moments give the BN backward terms in closed form.
//   * all cross-workgroup reductions are two-stage with fixed order (no float atomics): results are reproducible.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <limits.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"  // error codes
#include "../../include/liso_pillars.h"

namespace {

constexpr int kTile = 1024;          // points per tile in the rank pass
constexpr int kSlotEmpty = 0x7f7f7f7f;  // hipMemsetAsync(0x7f) sentinel; point indices are always smaller
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
                              int* __restrict__ first_enc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const float* p = pts + (size_t)i * C;
    const float cx = floorf((p[0] - cfg.x_min) / cfg.vx);
    const float cy = floorf((p[1] - cfg.y_min) / cfg.vy);
    const float cz = floorf((p[2] - cfg.z_min) / cfg.vz);
    // NaN fails every comparison below and is dropped
    const bool ok = cx >= 0.f && cx < (float)cfg.gx && cy >= 0.f && cy < (float)cfg.gy && cz >= 0.f && cz < 1.f;
    int cell = -1;
    if (ok) {
        const int b = sample_of(bi, batch, i);
        cell = (b * cfg.gx + (int)cx) * cfg.gy + (int)cy;
        atomicAdd(&count[cell], 1);
        atomicMax(&first_enc[cell], INT_MAX - i);  // == atomicMin over i with a zero-initialised array
    }
    cell_of_point[i] = cell;
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

// voxel_generator.py:275-278 keeps the first max_points points of a voxel: atomicMin cascade -> the max_points
// smallest indices, ascending, independent of arrival order.
__global__ void fill_kernel(const int* __restrict__ cell_of_point, const int* __restrict__ cell_to_voxel, int n_total,
                            int max_points, int* __restrict__ slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    const int cell = cell_of_point[i];
    if (cell < 0) return;
    const int v = cell_to_voxel[cell] - 1;
    if (v < 0) return;
    int* s = slots + (size_t)v * max_points;
    // the slot values only ever decrease: once the last slot holds a smaller index than ours we can never enter.
    // In crowded pillars (hundreds of points near the sensor) this removes almost all of the contended atomics.
    if (__hip_atomic_load(&s[max_points - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < i) return;
    int carry = i;
    for (int k = 0; k < max_points; k++) {
        const int old = atomicMin(&s[k], carry);
        if (old == kSlotEmpty) break;       // slot was empty: we filled it, nothing displaced
        carry = old > carry ? old : carry;  // keep the smaller in the slot, push the larger on
    }
}

// ---------------------------------------------------------------------------------------------------------
// PFN: shared row builder.  One wave per pillar; lane s < num owns point s.
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

// Builds the decorated rows of pillar v into frow[s][0..F) (and frow[s][F] = 1), returns num.
// Channel order (pillar_encoder.py:109-146 with legacy aliasing, :129-139): the in-place f_center update
// overwrites xyz of `features` itself, so the 10 inputs are [fc(3), extras(C-3), cluster(3), fc(3)], and because
// of the x/y swap at pcl_to_feature_grid.py:73, fc_x uses the *y index* and fc_y the *x index*.
template <int C>
__device__ __forceinline__ int build_rows(const float* __restrict__ pts, const int* __restrict__ slots,
                                          const int* __restrict__ coors, const int* __restrict__ num_points, int v,
                                          int max_points, const PfnGeom& g, float (*frow)[kFP], int lane) {
#pragma clang fp contract(off)
    constexpr int F = C + 6;
    const int num = num_points[v];
    const int xi = coors[v * 4 + 2], yi = coors[v * 4 + 3];
    float p[C];
#pragma unroll
    for (int k = 0; k < C; k++) p[k] = 0.f;
    if (lane < num) {
        const int idx = slots[(size_t)v * max_points + lane];
        const float* q = pts + (size_t)idx * C;
        if constexpr (C == 4) {
            const float4 t = *reinterpret_cast<const float4*>(q);
            p[0] = t.x; p[1] = t.y; p[2] = t.z; p[3] = t.w;
        } else {
#pragma unroll
            for (int k = 0; k < C; k++) p[k] = q[k];
        }
    }
    // points_mean = sum over the (zero padded) slots / num_points, pillar_encoder.py:112-115
    float sx = p[0], sy = p[1], sz = p[2];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        sz += __shfl_xor(sz, o);
    }
    const float fn = (float)num;
    const float mx = sx / fn, my = sy / fn, mz = sz / fn;
    if (lane < num) {
        const float fcx = p[0] - ((float)yi * g.vx + g.x_off);  // coors[:,3] (= y index) * vx, :130-132
        const float fcy = p[1] - ((float)xi * g.vy + g.y_off);  // coors[:,2] (= x index) * vy, :133-135
        const float fcz = p[2] - (0.f + g.z_off);               // coors[:,1] == 0,              :136-138
        float* f = frow[lane];
        f[0] = fcx; f[1] = fcy; f[2] = fcz;
#pragma unroll
        for (int k = 3; k < C; k++) f[k] = p[k];
        f[C + 0] = p[0] - mx; f[C + 1] = p[1] - my; f[C + 2] = p[2] - mz;
        f[C + 3] = fcx; f[C + 4] = fcy; f[C + 5] = fcz;
        f[F] = 1.f;
    }
    return num;
}

// ---- stats: second moments of the augmented feature vector [f, 1] over all valid rows ----------------------
template <int C>
__global__ __launch_bounds__(kPfnThreads) void pfn_stats_kernel(const float* __restrict__ pts, liso_pillar_cfg cfg,
                                                                int batch, const int* __restrict__ coors,
                                                                const int* __restrict__ num_points,
                                                                const int* __restrict__ slots,
                                                                const int* __restrict__ num_voxels,
                                                                double* __restrict__ partials) {
    constexpr int F = C + 6, D = F + 1, NP = D * (D + 1) / 2;
    __shared__ float frow[kPfnThreads / 64][kMaxPts][kFP];
    __shared__ double red[kPfnThreads / 64][LISO_PFN_STATS_DOUBLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PfnGeom g = pfn_geom(cfg);
    // lane -> up to two (j,k) pairs of the upper triangle
    int pj[2] = {0, 0}, pk[2] = {0, 0};
    {
        int e = 0;
        for (int j = 0; j < D; j++)
            for (int k = j; k < D; k++, e++) {
                if (e == lane) { pj[0] = j; pk[0] = k; }
                if (e == lane + 64) { pj[1] = j; pk[1] = k; }
            }
    }
    double acc0 = 0.0, acc1 = 0.0;
    const int rows = batch * cfg.max_voxels;
    for (int v = blockIdx.x * (kPfnThreads / 64) + wave; v < rows; v += gridDim.x * (kPfnThreads / 64)) {
        const int b = v / cfg.max_voxels;
        if (v - b * cfg.max_voxels >= num_voxels[b]) continue;
        const int num = build_rows<C>(pts, slots, coors, num_points, v, cfg.max_points, g, frow[wave], lane);
        __builtin_amdgcn_wave_barrier();
        for (int s = 0; s < num; s++) {
            const float* f = frow[wave][s];
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
    __shared__ double part[8][128];
    __shared__ double mom[LISO_PFN_STATS_DOUBLES];
    const int e = threadIdx.x & 127, chunk = threadIdx.x >> 7;  // 8 chunks of blocks
    double s = 0.0;
    if (e < LISO_PFN_STATS_DOUBLES) {
        const int per = (nblocks + 7) / 8;
        const int lo = chunk * per, hi = lo + per < nblocks ? lo + per : nblocks;
        for (int blk = lo; blk < hi; blk++) s += partials[(size_t)blk * LISO_PFN_STATS_DOUBLES + e];
    }
    part[chunk][e] = s;
    __syncthreads();
    if (threadIdx.x < LISO_PFN_STATS_DOUBLES) {
        double t = 0.0;
        for (int c = 0; c < 8; c++) t += part[c][threadIdx.x];
        mom[threadIdx.x] = t;
        moments[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x < kOut) {
        const int c = threadIdx.x;
        long long P = 0;
        for (int b = 0; b < batch; b++) P += num_voxels[b];
        const double M = (double)P * (double)max_points;  // BatchNorm1d sees [P, 64, 20]: padded rows count
        double w[F];
        for (int k = 0; k < F; k++) w[k] = (double)weight[c * F + k];
        double sum = 0.0, sq = 0.0;
        for (int j = 0; j < F; j++) {
            sum += w[j] * mom[pair_index(j, D - 1, D)];
            for (int k = 0; k < F; k++) {
                const int a = j <= k ? j : k, bb = j <= k ? k : j;
                sq += w[j] * w[k] * mom[pair_index(a, bb, D)];
            }
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

// ---- forward: Linear(F,64, no bias) -> BN -> ReLU -> max over the 20 slots -> canvas[b, x_idx, y_idx, :] ------------
// Dense-writer form: every canvas cell is written exactly once.
// A block owns 256 consecutive cells of the [B, gx, gy] grid.  Empty cells are zero-filled with 16-B stores (one
// coalesced 4-KiB wave-instruction per 32 bf16 cells), occupied cells get their 64 PFN channels; no separate memset
// pass over the 64*G^2 canvas and no double write.  HBM traffic = the canvas once + ~1 MB of cell->voxel indices.
constexpr int kCellsPerBlock = 256;

template <int C, typename OutT>
__global__ __launch_bounds__(kPfnThreads) void pfn_forward_dense_kernel(const float* __restrict__ pts, liso_pillar_cfg cfg,
                                                                        long total_cells, const int* __restrict__ coors,
                                                                        const int* __restrict__ num_points,
                                                                        const int* __restrict__ slots,
                                                                        const int* __restrict__ cell_to_voxel,
                                                                        const float* __restrict__ weight,
                                                                        const float* __restrict__ bn,
                                                                        OutT* __restrict__ canvas,
                                                                        float* __restrict__ occupancy) {
    constexpr int F = C + 6;
    constexpr int kChunks = kOut * (int)sizeof(OutT) / 16;  // 16-B chunks per cell (8 for bf16, 16 for fp32)
    __shared__ float frow[kPfnThreads / 64][kMaxPts][kFP];
    __shared__ int vox[kCellsPerBlock];
    __shared__ int list[kCellsPerBlock];
    __shared__ int cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long cell0 = (long)blockIdx.x * kCellsPerBlock;
    if (tid == 0) cnt = 0;
    __syncthreads();
    {
        const long cell = cell0 + tid;
        int v1 = 0;
        if (cell < total_cells) {
            v1 = cell_to_voxel[cell];
            occupancy[cell] = v1 > 0 ? 1.f : 0.f;
        }
        vox[tid] = v1;
        if (v1 > 0) list[atomicAdd(&cnt, 1)] = tid;
    }
    __syncthreads();
    // zero-fill the empty cells: consecutive threads -> consecutive 16-B chunks
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        uint4* base = reinterpret_cast<uint4*>(canvas + (size_t)cell0 * kOut);
        for (int id = tid; id < kCellsPerBlock * kChunks; id += kPfnThreads) {
            const int cl = id / kChunks;
            if (cell0 + cl < total_cells && vox[cl] == 0) base[id] = z;
        }
    }
    const int n_occ = cnt;
    if (n_occ == 0) return;
    const PfnGeom g = pfn_geom(cfg);
    float w[F];
#pragma unroll
    for (int k = 0; k < F; k++) w[k] = weight[lane * F + k];
    const float scale = bn[lane], shift = bn[kOut + lane];
    const float pad_val = fmaxf(shift, 0.f);
    for (int i = wave; i < n_occ; i += kPfnThreads / 64) {
        const int cl = list[i];
        const int v = vox[cl] - 1;
        const int num = build_rows<C>(pts, slots, coors, num_points, v, cfg.max_points, g, frow[wave], lane);
        __builtin_amdgcn_wave_barrier();
        float best = num < cfg.max_points ? pad_val : 0.f;
        for (int s = 0; s < num; s++) {
            const float* f = frow[wave][s];
            float x = 0.f;
#pragma unroll
            for (int k = 0; k < F; k++) x = fmaf(w[k], f[k], x);
            best = fmaxf(best, fmaxf(fmaf(x, scale, shift), 0.f));
        }
        __builtin_amdgcn_wave_barrier();
        store_out<OutT>(canvas + ((size_t)cell0 + cl) * kOut + lane, best);
    }
}

// ---- backward: per-block partial sums of  A[c][k] = sum dz*f,  dbeta[c] = sum dz,  dgamma[c] = sum dz*xhat ------
template <int C, typename GT>
__global__ __launch_bounds__(kPfnThreads) void pfn_backward_kernel(const float* __restrict__ pts, liso_pillar_cfg cfg,
                                                                   int batch, const int* __restrict__ coors,
                                                                   const int* __restrict__ num_points,
                                                                   const int* __restrict__ slots,
                                                                   const int* __restrict__ num_voxels,
                                                                   const float* __restrict__ weight,
                                                                   const float* __restrict__ bn,
                                                                   const GT* __restrict__ grad_canvas,
                                                                   float* __restrict__ partials) {
    constexpr int F = C + 6, NA = F + 2;
    __shared__ float frow[kPfnThreads / 64][kMaxPts][kFP];
    __shared__ float red[kPfnThreads / 64][NA][kOut];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PfnGeom g = pfn_geom(cfg);
    float w[F], acc[NA];
#pragma unroll
    for (int k = 0; k < F; k++) w[k] = weight[lane * F + k];
#pragma unroll
    for (int k = 0; k < NA; k++) acc[k] = 0.f;
    const float scale = bn[lane], shift = bn[kOut + lane], mean = bn[2 * kOut + lane], invstd = bn[3 * kOut + lane];
    const float pad_val = fmaxf(shift, 0.f);
    const int rows = batch * cfg.max_voxels;
    for (int v = blockIdx.x * (kPfnThreads / 64) + wave; v < rows; v += gridDim.x * (kPfnThreads / 64)) {
        const int b = v / cfg.max_voxels;
        if (v - b * cfg.max_voxels >= num_voxels[b]) continue;
        const int num = build_rows<C>(pts, slots, coors, num_points, v, cfg.max_points, g, frow[wave], lane);
        __builtin_amdgcn_wave_barrier();
        float best = -1.f, best_x = 0.f;
        int best_s = -1;
        for (int s = 0; s < num; s++) {
            const float* f = frow[wave][s];
            float x = 0.f;
#pragma unroll
            for (int k = 0; k < F; k++) x = fmaf(w[k], f[k], x);
            const float y = fmaxf(fmaf(x, scale, shift), 0.f);
            if (y > best) { best = y; best_s = s; best_x = x; }
        }
        if (num < cfg.max_points && pad_val > best) { best = pad_val; best_s = -1; best_x = 0.f; }
        const int xi = coors[v * 4 + 2], yi = coors[v * 4 + 3];
        const size_t cellidx = ((size_t)b * cfg.gx + xi) * cfg.gy + yi;
        const float gy = load_in<GT>(grad_canvas + cellidx * kOut + lane);
        const float dz = best > 0.f ? gy : 0.f;  // ReLU gate; the max routes the gradient to one row
        if (best_s >= 0) {
            const float* f = frow[wave][best_s];
#pragma unroll
            for (int k = 0; k < F; k++) acc[k] = fmaf(dz, f[k], acc[k]);
        }
        acc[F] += dz;
        acc[F + 1] = fmaf(dz, (best_x - mean) * invstd, acc[F + 1]);
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
    for (int blk = lo; blk < hi; blk++) s += (double)partials[((size_t)blk * NA + k) * kOut + c];
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

template <int C>
void launch_bn_prepare(const float* points, const liso_pillar_cfg* cfg, int batch, const int* coors,
                       const int* num_points, const int* slots, const int* num_voxels, const float* weight,
                       const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                       float eps, float* bn_out, double* moments, double* partials, int grid, hipStream_t st) {
    pfn_stats_kernel<C><<<grid, kPfnThreads, 0, st>>>(points, *cfg, batch, coors, num_points, slots, num_voxels, partials);
    pfn_bn_finalize_kernel<C><<<1, 1024, 0, st>>>(partials, grid, num_voxels, batch, cfg->max_points, weight, gamma, beta,
                                                  running_mean, running_var, momentum, eps, bn_out, moments);
}

}  // namespace

extern "C" {

size_t liso_pillars_voxelize_workspace_bytes(const liso_pillar_cfg* cfg, int batch, int n_total) {
    if (!cfg_ok(cfg, batch) || n_total < 0) return 0;
    const size_t cells = (size_t)batch * cfg->gx * cfg->gy;
    const size_t tiles = (size_t)(n_total + kTile - 1) / kTile + batch;
    return (2 * cells + (size_t)n_total + tiles + 64) * sizeof(int);
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
    if (workspace_bytes < liso_pillars_voxelize_workspace_bytes(cfg, batch, n_total)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)batch * cfg->gx * cfg->gy;
    int* count = (int*)workspace;
    int* first_enc = count + cells;
    int* cell_of_point = first_enc + cells;
    int* tile_count = cell_of_point + n_total;
    const BatchInfo bi = make_batch(offsets_host, batch);
    const int tiles = bi.tile_off[batch];
    if (hipMemsetAsync(count, 0, 2 * cells * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (hipMemsetAsync(cell_to_voxel, 0, cells * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (hipMemsetAsync(num_voxels, 0, batch * sizeof(int), st) != hipSuccess) return LISO_ELAUNCH;
    if (hipMemsetAsync(slots, 0x7f, (size_t)batch * cfg->max_voxels * cfg->max_points * sizeof(int), st) != hipSuccess)
        return LISO_ELAUNCH;
    if (n_total == 0) return LISO_OK;
    const int nb = (n_total + 255) / 256;
    hipLaunchKernelGGL(assign_kernel, dim3(nb), dim3(256), 0, st, points, n_total, cfg->n_channels, bi, batch, *cfg,
                       cell_of_point, count, first_enc);
    hipLaunchKernelGGL(tile_count_kernel, dim3(tiles), dim3(kTile), 0, st, cell_of_point, first_enc, bi, batch,
                       tile_count);
    hipLaunchKernelGGL(rank_kernel, dim3(tiles), dim3(kTile), 0, st, cell_of_point, first_enc, count, bi, batch, *cfg,
                       tile_count, coors, num_points, cell_to_voxel, num_voxels);
    hipLaunchKernelGGL(fill_kernel, dim3(nb), dim3(256), 0, st, cell_of_point, cell_to_voxel, n_total, cfg->max_points,
                       slots);
    return check_launch();
}

size_t liso_pfn_partials_bytes(void) {
    const size_t stats = (size_t)kPfnGrid * LISO_PFN_STATS_DOUBLES * sizeof(double);
    const size_t bwd = (size_t)kPfnGrid * (11 + 2) * kOut * sizeof(float) + (11 + 2) * kOut * sizeof(double);
    return stats > bwd ? stats : bwd;
}

int liso_pfn_bn_prepare_f32(const float* points, const liso_pillar_cfg* cfg, int batch, const int* coors,
                            const int* num_points, const int* slots, const int* num_voxels, const float* weight,
                            const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, int training, float* bn_out, double* moments, void* partials,
                            void* stream) {
    if (!cfg_ok(cfg, batch) || !gamma || !beta || !running_mean || !running_var || !bn_out) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!training) {
        pfn_bn_eval_kernel<<<1, 64, 0, st>>>(gamma, beta, running_mean, running_var, eps, bn_out);
        return check_launch();
    }
    if (!points || !coors || !num_points || !slots || !num_voxels || !weight || !moments || !partials)
        return LISO_EINVAL;
    const int grid = pfn_grid(batch * cfg->max_voxels);
    switch (cfg->n_channels) {
        case 3: launch_bn_prepare<3>(points, cfg, batch, coors, num_points, slots, num_voxels, weight, gamma, beta, running_mean, running_var, momentum, eps, bn_out, moments, (double*)partials, grid, st); break;
        case 4: launch_bn_prepare<4>(points, cfg, batch, coors, num_points, slots, num_voxels, weight, gamma, beta, running_mean, running_var, momentum, eps, bn_out, moments, (double*)partials, grid, st); break;
        default: launch_bn_prepare<5>(points, cfg, batch, coors, num_points, slots, num_voxels, weight, gamma, beta, running_mean, running_var, momentum, eps, bn_out, moments, (double*)partials, grid, st); break;
    }
    return check_launch();
}

int liso_pfn_forward_scatter(const float* points, const liso_pillar_cfg* cfg, int batch, const int* coors,
                             const int* num_points, const int* slots, const int* cell_to_voxel, const float* weight,
                             const float* bn_out, void* canvas, int out_bf16, float* occupancy, void* stream) {
    if (!cfg_ok(cfg, batch) || !points || !coors || !num_points || !slots || !cell_to_voxel || !weight || !bn_out ||
        !canvas || !occupancy)
        return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long cells = (long)batch * cfg->gx * cfg->gy;
    const unsigned grid = (unsigned)((cells + kCellsPerBlock - 1) / kCellsPerBlock);
#define LISO_FWD(CC, T) pfn_forward_dense_kernel<CC, T><<<grid, kPfnThreads, 0, st>>>(points, *cfg, cells, coors, num_points, slots, cell_to_voxel, weight, bn_out, (T*)canvas, occupancy)
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

int liso_pfn_backward(const float* points, const liso_pillar_cfg* cfg, int batch, const int* coors,
                      const int* num_points, const int* slots, const int* num_voxels, const float* weight,
                      const float* gamma, const float* bn_out, const double* moments, int training,
                      const void* grad_canvas, int grad_bf16, float* grad_weight, float* grad_gamma,
                      float* grad_beta, void* partials, void* stream) {
    if (!cfg_ok(cfg, batch) || !points || !coors || !num_points || !slots || !num_voxels || !weight || !gamma ||
        !bn_out || !grad_canvas || !grad_weight || !grad_gamma || !grad_beta || !partials || (training && !moments))
        return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = pfn_grid(batch * cfg->max_voxels);
#define LISO_BWD(CC, T)                                                                                              \
    do {                                                                                                             \
        constexpr int NA_ = CC + 8;                                                                                  \
        double* tot_ = (double*)((float*)partials + (size_t)kPfnGrid * (11 + 2) * kOut);                             \
        pfn_backward_kernel<CC, T><<<grid, kPfnThreads, 0, st>>>(points, *cfg, batch, coors, num_points, slots,       \
                                                                 num_voxels, weight, bn_out, (const T*)grad_canvas,  \
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
