// BEV grid -> per-point gather and its deterministic adjoint for gfx950.  C ABI + reference lines: include/liso_slim.h.
//
//   bev_gather_fwd   one thread per (point, channel): out[n, c] = lin[n] < 0 ? default : grid[lin[n], c]
//                    (channels-last grid rows of C floats; writes coalesced, reads one <= 128-B row per point)
//   bev_gather_bwd   points pre-sorted by cell (host: one sort per cloud, reused by every decode of the step); a
//                    two-pass segmented sum writes every occupied cell's gradient row once: no atomics, fixed
//                    summation order, bit reproducible.  Cells without points keep the caller's zeros.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

__global__ void bev_gather_fwd_kernel(const float* __restrict__ grid, const int* __restrict__ lin, long n_rows, int c,
                                      float default_value, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * c) return;
    const long n = i / c;
    const int ch = (int)(i - n * c);
    const int l = lin[n];
    out[i] = l < 0 ? default_value : grid[(size_t)l * c + ch];
}

// Long segments (a wall puts hundreds of points into one pillar) are cut into chunks of kChunk rows: pass 1 sums every
// chunk (<= kChunk serial steps per thread) into `partial` at the chunk head's row, pass 2 lets every segment head add
// up its chunk heads (<= len / kChunk steps) and write the cell's gradient row.  Fixed order, no atomics.
constexpr int kChunk = 16;

// LANES lanes per sorted row (32; maps of <= 8 channels take the row-per-thread kernels below): the row's cell / rank are read once per
// lane group (not once per channel), lanes = channels.
template <int LANES>
__global__ __launch_bounds__(256) void bev_gather_bwd_chunk_kernel(const float* __restrict__ grad_out, const int* __restrict__ sorted_lin,
                                                                   const int* __restrict__ order, const int* __restrict__ seg_rank,
                                                                   long n_rows, int c, float* __restrict__ partial) {
    const long s = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const int lane = threadIdx.x % LANES;
    if (s >= n_rows) return;
    const int cell = sorted_lin[s];
    const int rank0 = seg_rank[s];
    if (cell < 0 || (rank0 % kChunk) != 0) return;
    // rows of this chunk: consecutive sorted rows of the same cell, at most kChunk.  Row s + k belongs to it iff its rank
    // inside the cell's run is rank0 + k: all kChunk ranks and source rows are fetched up front (independent loads), so
    // the chunk costs two memory round trips instead of two per row.
    int src[kChunk];
    int rows = kChunk;
#pragma unroll
    for (int k = 0; k < kChunk; k++) {
        const bool in_range = s + k < n_rows;
        const int rk = in_range ? seg_rank[s + k] : -1;
        src[k] = in_range ? order[s + k] : 0;
        if (rows == kChunk && rk != rank0 + k) rows = k;
    }
    for (int ch = lane; ch < c; ch += LANES) {
        float v[kChunk];
#pragma unroll
        for (int k = 0; k < kChunk; k++) v[k] = k < rows ? grad_out[(size_t)src[k] * c + ch] : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < kChunk; k++) acc += v[k];  // fixed order; rows beyond the chunk add exact zeros
        partial[(size_t)s * c + ch] = acc;
    }
}

template <int LANES>
__global__ __launch_bounds__(256) void bev_gather_bwd_segment_kernel(const float* __restrict__ partial, const int* __restrict__ sorted_lin,
                                                                     const int* __restrict__ seg_rank, long n_rows, int c,
                                                                     float* __restrict__ grad_grid) {
    const long s = ((long)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const int lane = threadIdx.x % LANES;
    if (s >= n_rows) return;
    const int cell = sorted_lin[s];
    if (cell < 0 || seg_rank[s] != 0) return;
    for (int ch = lane; ch < c; ch += LANES) {
        float acc = 0.f;
        for (long j = s; j < n_rows && sorted_lin[j] == cell; j += kChunk) acc += partial[(size_t)j * c + ch];
        grad_grid[(size_t)cell * c + ch] = acc;
    }
}

// ---- maps of <= 8 channels (the decoder's 8-channel network output, the 3-channel flow | weight maps): a thread per sorted row ----
// The chunk / segment pair above keeps one lane group per sorted row, of which only the chunk heads (one row in ~10) do anything, each
// through a chain of dependent global loads: 189 + 106 us for the 1.44 M rows of a SLIM step (0.3 TB/s).  Here EVERY thread gathers its
// row (<= 32 bytes) in one round trip, and the sums run in LDS:
//   rows kernel      block = 256 consecutive sorted rows.  Level 1: rows whose index is a multiple of 16 or that start a run of equal
//                    cells sum forward to the next such row; level 2: run starts add the level-1 sums of their run (<= 16 + 16 steps, all
//                    in LDS).  A run that starts (rank 0) and ends inside the block writes its cell's gradient row; otherwise the
//                    block's share goes to partial[first row of the share].
//   boundary kernel  a thread per block boundary: the run that crosses it FIRST there (its start lies in the block in front) adds the
//                    shares of all blocks it touches, in block order, and writes the cell's row.
// Fixed order, no atomics: bit reproducible.
constexpr int kRowsBlock = 256;

__global__ __launch_bounds__(kRowsBlock) void bev_gather_bwd_rows_kernel(const float* __restrict__ grad_out, const int* __restrict__ sorted_lin,
                                                                         const int* __restrict__ order, const int* __restrict__ seg_rank,
                                                                         long n_rows, int c, float* __restrict__ partial,
                                                                         float* __restrict__ grad_grid) {
    __shared__ float val[kRowsBlock][9];  // (row stride 9: neighbouring rows of one channel fall into different banks)
    __shared__ int cells[kRowsBlock + 1];
    const int tid = threadIdx.x;
    const long s = (long)blockIdx.x * kRowsBlock + tid;
    int cell = -1;
    float v[8];
#pragma unroll
    for (int ch = 0; ch < 8; ch++) v[ch] = 0.f;
    if (s < n_rows) {
        cell = sorted_lin[s];
        if (cell >= 0) {
            const float* g = grad_out + (size_t)order[s] * c;
            if (c == 8) {
                const float4 a = *reinterpret_cast<const float4*>(g), b = *reinterpret_cast<const float4*>(g + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int ch = 0; ch < 8; ch++)
                    if (ch < c) v[ch] = g[ch];
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 8; ch++) val[tid][ch] = v[ch];
    cells[tid] = cell;
    if (tid == kRowsBlock - 1) cells[kRowsBlock] = s + 1 < n_rows ? sorted_lin[s + 1] : -2;
    __syncthreads();
    const bool head = cell >= 0 && (tid == 0 || cells[tid - 1] != cell);
    if (cell >= 0 && (head || (tid & 15) == 0)) {  // level 1: forward to the next multiple of 16 / the end of the run
        for (int k = tid + 1; k < kRowsBlock && (k & 15) != 0 && cells[k] == cell; k++)
#pragma unroll
            for (int ch = 0; ch < 8; ch++) v[ch] += val[k][ch];
    }
    __syncthreads();  // (level 1 only read rows that nobody overwrites: a row is written by its own thread)
    if (cell >= 0 && (head || (tid & 15) == 0))
#pragma unroll
        for (int ch = 0; ch < 8; ch++) val[tid][ch] = v[ch];
    __syncthreads();
    if (!head) return;
    for (int k = (tid | 15) + 1; k < kRowsBlock && cells[k] == cell; k += 16)  // level 2: the run's level-1 sums, in row order
#pragma unroll
        for (int ch = 0; ch < 8; ch++) v[ch] += val[k][ch];
    const bool ends_here = !(cells[kRowsBlock - 1] == cell && cells[kRowsBlock] == cell);
    float* dst = (ends_here && seg_rank[s] == 0) ? grad_grid + (size_t)cell * c : partial + (size_t)s * c;
#pragma unroll
    for (int ch = 0; ch < 8; ch++)
        if (ch < c) dst[ch] = v[ch];
}

__global__ __launch_bounds__(256) void bev_gather_bwd_boundary_kernel(const float* __restrict__ partial, const int* __restrict__ sorted_lin,
                                                                      const int* __restrict__ seg_rank, long n_rows, int c,
                                                                      float* __restrict__ grad_grid) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x + 1;  // boundary in front of block b
    const long r = b * kRowsBlock;
    if (r >= n_rows) return;
    const int cell = sorted_lin[r];
    if (cell < 0) return;
    const int rank = seg_rank[r];
    const long h = r - rank;  // where the run starts
    if (rank == 0 || h / kRowsBlock != b - 1) return;  // no run crosses here, or it crossed an earlier boundary first
    float v[8];
#pragma unroll
    for (int ch = 0; ch < 8; ch++) v[ch] = ch < c ? partial[(size_t)h * c + ch] : 0.f;
    for (long q = r; q < n_rows && sorted_lin[q] == cell; q += kRowsBlock)
#pragma unroll
        for (int ch = 0; ch < 8; ch++)
            if (ch < c) v[ch] += partial[(size_t)q * c + ch];
    for (int ch = 0; ch < c; ch++) grad_grid[(size_t)cell * c + ch] = v[ch];
}

// BevGatherPlan's list (liso/slim/slim_loss/static_aggregation.py:69-84 indexes grid[b, row, col] per valid point): the flattened
// cell of every point in ONE launch instead of 9 framework launches (.long(), arange, 2 x (mul, add), where + its scalar tensor, .to(int32))
template <typename T>
__global__ __launch_bounds__(256) void bev_lin_index_kernel(const T* __restrict__ coors, const unsigned char* __restrict__ valid, long n_rows,
                                                            long n, int h, int w, int* __restrict__ lin) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const long b = i / n;
    const long v = (b * h + (long)coors[i * 2 + 0]) * w + (long)coors[i * 2 + 1];  // (int64 like the reference's .long() arithmetic)
    lin[i] = valid[i] ? (int)v : -1;
}

// ---- the plan's index arithmetic as three launches (BevGatherPlan.tiled / _sort: 48 framework launches per SLIM step before) ----------
// copy j of the tiled batch [samples[:half]] * n_it + [samples[half:]] * n_it shows distinct sample src(j); its cells are shifted by
// (j - src) * H * W
__device__ __forceinline__ int tiled_src(int j, int n2, int n_it, int half) {
    return j < half * n_it ? j % half : half + (j - half * n_it) % (n2 - half);
}

__global__ __launch_bounds__(256) void bev_plan_tile_lin_kernel(const int* __restrict__ rows, int n2, long n, int n_it, int half, int hw,
                                                                int* __restrict__ lin) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n2 * n_it * n) return;
    const int j = (int)(i / n);
    const long k = i - (long)j * n;
    const int src = tiled_src(j, n2, n_it, half);
    const int v = rows[(long)src * n + k];
    lin[i] = v >= 0 ? v + (j - src) * hw : v;
}

// first index whose value is >= key in the ascending list s[0, n)
__device__ __forceinline__ long lower_bound(const int* __restrict__ s, long n, int key) {
    long lo = 0, hi = n;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (s[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// rank of every row inside its run of equal cells = position - first position of the value; `order` narrowed to int32 on the way
__global__ __launch_bounds__(256) void bev_plan_rank_kernel(const int* __restrict__ sorted_lin, const long long* __restrict__ order64, long n,
                                                            int* __restrict__ rank, int* __restrict__ order32) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rank[i] = (int)(i - lower_bound(sorted_lin, n, sorted_lin[i]));
    if (order32) order32[i] = (int)order64[i];
}

// the sorted view of the tiled batch from the flat sort of the distinct samples ([invalid rows | sample 0's cells | sample 1's cells | ...]):
// copy j's block = its sample's valid rows in sorted order, cells shifted, then invalid padding up to n rows.  One block = 256 rows of one copy.
__global__ __launch_bounds__(256) void bev_plan_expand_kernel(const int* __restrict__ s_flat, const long long* __restrict__ o_flat,
                                                              const int* __restrict__ rank_flat, int n2, long n, int n_it, int half, int hw,
                                                              int* __restrict__ sorted_lin, int* __restrict__ order, int* __restrict__ rank) {
    __shared__ long se[2];
    const int j = blockIdx.y;
    const int src = tiled_src(j, n2, n_it, half);
    if (threadIdx.x < 2) se[threadIdx.x] = lower_bound(s_flat, (long)n2 * n, (src + (int)threadIdx.x) * hw);
    __syncthreads();
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const long start = se[0], nv = se[1] - se[0];
    const long o = (long)j * n + k;
    if (k < nv) {
        const long idx = start + k;
        sorted_lin[o] = s_flat[idx] + (j - src) * hw;
        order[o] = (int)(o_flat[idx] - (long)src * n + (long)j * n);
        rank[o] = rank_flat[idx];
    } else {
        sorted_lin[o] = -1;
        order[o] = 0;
        rank[o] = 0;
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

}  // namespace

extern "C" {

int liso_bev_gather_fwd_f32(const float* grid, const int* lin, long n_rows, int c, float default_value, float* out,
                            void* stream) {
    if (n_rows < 0 || c < 1) return LISO_EINVAL;
    if (n_rows == 0) return LISO_OK;
    if (!grid || !lin || !out) return LISO_EINVAL;
    const long total = n_rows * c;
    bev_gather_fwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(grid, lin, n_rows, c, default_value, out);
    return check_launch();
}

int liso_bev_lin_index(const void* coors, int coors_are_int64, const unsigned char* valid, int batch, long n, int h, int w, int* lin,
                       void* stream) {
    if (batch < 0 || n < 0 || h <= 0 || w <= 0) return LISO_EINVAL;
    const long rows = (long)batch * n;
    if (rows == 0) return LISO_OK;
    if (!coors || !valid || !lin) return LISO_EINVAL;
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    if (coors_are_int64)
        bev_lin_index_kernel<long long><<<blocks, 256, 0, (hipStream_t)stream>>>((const long long*)coors, valid, rows, n, h, w, lin);
    else
        bev_lin_index_kernel<int><<<blocks, 256, 0, (hipStream_t)stream>>>((const int*)coors, valid, rows, n, h, w, lin);
    return check_launch();
}

int liso_bev_plan_tile_lin(const int* rows, int n2, long n, int n_it, int half, int h, int w, int* lin, void* stream) {
    if (n2 <= 0 || n < 0 || n_it < 1 || half <= 0 || half > n2 || h <= 0 || w <= 0) return LISO_EINVAL;
    const long total = (long)n2 * n_it * n;
    if (total == 0) return LISO_OK;
    if (!rows || !lin) return LISO_EINVAL;
    bev_plan_tile_lin_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(rows, n2, n, n_it, half, h * w, lin);
    return check_launch();
}

int liso_bev_plan_rank(const int* sorted_lin, const long long* order64, long n, int* rank, int* order32, void* stream) {
    if (n < 0) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    if (!sorted_lin || !rank || (order32 && !order64)) return LISO_EINVAL;
    bev_plan_rank_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(sorted_lin, order64, n, rank, order32);
    return check_launch();
}

int liso_bev_plan_expand(const int* s_flat, const long long* o_flat, const int* rank_flat, int n2, long n, int n_it, int half, int h, int w,
                         int* sorted_lin, int* order, int* rank, void* stream) {
    if (n2 <= 0 || n < 0 || n_it < 1 || half <= 0 || half > n2 || h <= 0 || w <= 0 || (long)n2 * h * w > 0x7fffffffL) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    if (!s_flat || !o_flat || !rank_flat || !sorted_lin || !order || !rank) return LISO_EINVAL;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)(n2 * n_it));
    bev_plan_expand_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(s_flat, o_flat, rank_flat, n2, n, n_it, half, h * w, sorted_lin, order, rank);
    return check_launch();
}

int liso_bev_gather_bwd_f32(const float* grad_out, const int* sorted_lin, const int* order, const int* seg_rank, long n_rows,
                            int c, float* partial, float* grad_grid, void* stream) {
    if (n_rows < 0 || c < 1) return LISO_EINVAL;
    if (n_rows == 0) return LISO_OK;
    if (!grad_out || !sorted_lin || !order || !seg_rank || !partial || !grad_grid) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (c <= 8) {
        const long nblk = (n_rows + kRowsBlock - 1) / kRowsBlock;
        bev_gather_bwd_rows_kernel<<<(unsigned)nblk, kRowsBlock, 0, st>>>(grad_out, sorted_lin, order, seg_rank, n_rows, c, partial, grad_grid);
        if (nblk > 1)
            bev_gather_bwd_boundary_kernel<<<(unsigned)((nblk - 1 + 255) / 256), 256, 0, st>>>(partial, sorted_lin, seg_rank, n_rows, c, grad_grid);
    } else {
        const unsigned blocks = (unsigned)((n_rows * 32 + 255) / 256);
        bev_gather_bwd_chunk_kernel<32><<<blocks, 256, 0, st>>>(grad_out, sorted_lin, order, seg_rank, n_rows, c, partial);
        bev_gather_bwd_segment_kernel<32><<<blocks, 256, 0, st>>>(partial, sorted_lin, seg_rank, n_rows, c, grad_grid);
    }
    return check_launch();
}

}  // extern "C"
