// Sparse form of the SLIM encoders' first convolution (7x7, stride 2, padding 3, 64 -> 32 channels, fp32 tensors) for gfx950.
// C ABI + the reference lines it replaces: include/liso_conv.h (liso_sparse_stem_*).
//
// The input is the pillar canvas: 1-6 % of its 512^2 cells hold a pillar, the rest are exact zeros.  The dense implicit GEMM (with
// tile skipping) still multiplies every 8 x 32-pixel tile that has one occupied cell in its window: 0.39 ms for 8 sweeps, 500 MB
// read.  Here only occupied cells are multiplied:
//   1. cells_rows / cells_scan / cells_fill: the occupied cells of every canvas row, split by the PARITY of (row, column) -- a
//      stride-2 convolution reaches a cell through the kernel taps of one parity class only (4 x 4, 4 x 3, 3 x 4 or 3 x 3 of the 49)
//      -- listed per class in (sample, row, column) order, class segments aligned to 128 entries; an occupancy bitmap and the
//      cell -> list position map on the side.
//   2. stem_taps_kernel: for 128 cells of one class per block, the products of the cell's 64 features with the filters of every tap
//      of its class: [128 x 64] . [64 x 32] per tap on the matrix cores in F32X3 arithmetic (fp32 features split into bf16 hi + lo in
//      registers, the convolution kernels' packed weight panels as B operands), written as rows of 16 tap slots x 32 channels.
//   3. stem_gather_kernel: every OUTPUT pixel adds the tap products of the occupied cells in its 7 x 7 window in (ky, kx) order (the
//      bitmap says which, the position map where), plus bias; ReLU or the per-block statistics partial sums of the pending
//      InstanceNorm as epilogue.  Every output pixel is written exactly once; no atomics, fixed summation order.
// Cost: products of ~4k cells per sweep instead of 65k output pixels x 49 taps; the dense output (8.4 MB per sweep) is the floor.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/liso_conv.h"
#include "../../include/liso_iou3d.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int CI = 64, CO = 32, KS = 7, TAPS = 49, PAD = 3;
constexpr int SLOTS = 16;            // tap slots per cell row of the product buffer: (ky >> 1) * 4 + (kx >> 1)
constexpr int ROWF = SLOTS * CO;     // floats per cell in the product buffer
constexpr int NP = 64, KP8 = CI / 8; // packed panels: [plane][tap][CI / 8][NP][8] bf16 (liso_conv_pack_weights, F32X3)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float round_bf16(float v) { return (float)(__bf16)v; }
__device__ __forceinline__ bf8 as_bf8(const uint4& v) { return __builtin_bit_cast(bf8, v); }

// class of a cell = ((row + 1) & 1) * 2 + ((column + 1) & 1): the parity of the kernel rows / columns that reach it
// ---- 1a. per canvas row: occupied cells with even / odd column, occupancy bitmap ----------------------------------------------------
__global__ __launch_bounds__(64) void cells_rows_kernel(const float* __restrict__ occ, int wi, int words, int n_rows,
                                                        int* __restrict__ cnt4, unsigned* __restrict__ bitmap) {
    const long row = blockIdx.x;
    const int lane = threadIdx.x;
    int even = 0, odd = 0;
    for (int c0 = 0; c0 < wi; c0 += 64) {
        const int c = c0 + lane;
        const bool on = c < wi && occ[row * wi + c] != 0.0f;
        const unsigned long long m = __ballot(on);
        even += __popcll(m & 0x5555555555555555ull);  // (c0 is a multiple of 64: lane parity = column parity)
        odd += __popcll(m & 0xaaaaaaaaaaaaaaaaull);
        if (lane == 0 && (c0 >> 5) < words) bitmap[row * words + (c0 >> 5)] = (unsigned)m;
        if (lane == 32 && (c0 >> 5) + 1 < words) bitmap[row * words + (c0 >> 5) + 1] = (unsigned)(m >> 32);
    }
    if (lane == 0) {
        // cnt4[class][row], class = cy * 2 + cx with cy = (row + 1) & 1 (canvas heights are even: the global row index has the row's
        // parity inside its sample) and cx = (column + 1) & 1: even columns are cx = 1, odd columns cx = 0
        const int cy = (int)((row + 1) & 1);
        cnt4[(cy * 2 + 0) * n_rows + row] = odd;
        cnt4[(cy * 2 + 1) * n_rows + row] = even;
        cnt4[((1 - cy) * 2 + 0) * n_rows + row] = 0;
        cnt4[((1 - cy) * 2 + 1) * n_rows + row] = 0;
    }
}

// ---- 1b. one block: exclusive scan of cnt4 in (class, row) order, class segments aligned to 128; seg[0..3] = class bases,
//          seg[4..7] = class ends, seg[8] = 1 if the capacity was exceeded -----------------------------------------------------------
__global__ __launch_bounds__(1024) void cells_scan_kernel(const int* __restrict__ cnt4, int n_rows, int cap, int* __restrict__ off4,
                                                          int* __restrict__ seg, int* __restrict__ overflow) {
    __shared__ int part[1024];
    __shared__ int cls_total[4];
    const int tid = threadIdx.x;
    for (int cls = 0; cls < 4; cls++) {  // (four scans of n_rows entries: 6k-12k rows, a few microseconds)
        const int* c = cnt4 + (size_t)cls * n_rows;
        const int per = (n_rows + 1023) / 1024;
        const int b0 = tid * per;
        int s = 0;
        for (int i = 0; i < per; i++)
            if (b0 + i < n_rows) s += c[b0 + i];
        __syncthreads();
        part[tid] = s;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int v = tid >= o ? part[tid - o] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        int run = tid > 0 ? part[tid - 1] : 0;
        for (int i = 0; i < per; i++)
            if (b0 + i < n_rows) {
                off4[(size_t)cls * n_rows + b0 + i] = run;  // relative to the class base (added by the consumers)
                run += c[b0 + i];
            }
        if (tid == 1023) cls_total[cls] = part[1023];
        __syncthreads();
    }
    if (tid == 0) {
        int base = 0, over = 0;
        for (int cls = 0; cls < 4; cls++) {
            int end = base + cls_total[cls];
            if (end > cap) { end = cap > base ? cap : base; over = 1; }
            seg[cls] = base;
            seg[4 + cls] = end;
            base = (end + 127) / 128 * 128;
        }
        seg[8] = over;
        if (over && overflow) *overflow = 1;  // (sticky: the caller zeroes it once and looks at it whenever it likes)
    }
}

// ---- 1c. per canvas row: list position of every occupied cell (class base + row offset + rank among the row's cells of its column
//          parity), cell index into the list, position into the map ------------------------------------------------------------------
__global__ __launch_bounds__(64) void cells_fill_kernel(const float* __restrict__ occ, int wi, int n_rows, const int* __restrict__ off4,
                                                        const int* __restrict__ seg, int* __restrict__ cells, int* __restrict__ cell_pos) {
    const long row = blockIdx.x;
    const int lane = threadIdx.x;
    const int cy = (int)((row + 1) & 1);
    // cx = 0: odd columns, cx = 1: even columns
    int base[2] = {seg[cy * 2 + 0] + off4[(size_t)(cy * 2 + 0) * n_rows + row], seg[cy * 2 + 1] + off4[(size_t)(cy * 2 + 1) * n_rows + row]};
    const int end[2] = {seg[4 + cy * 2 + 0], seg[4 + cy * 2 + 1]};
    for (int c0 = 0; c0 < wi; c0 += 64) {
        const int c = c0 + lane;
        const bool on = c < wi && occ[row * wi + c] != 0.0f;
        const unsigned long long m = __ballot(on);
        const unsigned long long me = m & 0x5555555555555555ull, mo = m & 0xaaaaaaaaaaaaaaaaull;
        if (on) {
            const int cx = (lane & 1) ? 0 : 1;
            const unsigned long long mine = (lane & 1) ? mo : me;
            const int pos = base[cx] + __popcll(mine & ((1ull << lane) - 1ull));
            if (pos < end[cx]) {
                cells[pos] = (int)(row * wi + c);
                cell_pos[row * wi + c] = pos;
            } else {
                cell_pos[row * wi + c] = -1;  // beyond the capacity (seg[8] is set): the cell is dropped
            }
        }
        base[0] += __popcll(mo);
        base[1] += __popcll(me);
    }
}

// ---- 2. tap products of 128 cells of one class per block --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_taps_kernel(const float* __restrict__ x, long xps, const int* __restrict__ cells,
                                                        const int* __restrict__ seg, const uint4* __restrict__ wp,
                                                        float* __restrict__ prod) {
    const int pos0 = blockIdx.x * 128;
    int cls = -1;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (pos0 >= seg[q] && pos0 < seg[4 + q]) cls = q;
    if (cls < 0) return;
    const int end = seg[4 + cls];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = pos0 + wave * 32;
    if (row0 >= end) return;
    // A fragments: the features of cell row0 + r, channels ks * 16 + h * 8 .. + 7, as bf16 hi / lo
    uint4 ah[4], al[4];
    {
        const int p = row0 + r;
        const bool ok = p < end;
        const float* src = x + (size_t)(ok ? cells[p] : cells[row0]) * xps;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + ks * 16 + h * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(src + ks * 16 + h * 8 + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            unsigned hi2[4], lo2[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
                hi2[e] = pack_bf16(h0, h1);
                lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
            }
            ah[ks] = make_uint4(hi2[0], hi2[1], hi2[2], hi2[3]);
            al[ks] = make_uint4(lo2[0], lo2[1], lo2[2], lo2[3]);
        }
    }
    const int cy = cls >> 1, cx = cls & 1;
    for (int ky = cy; ky < KS; ky += 2) {
        for (int kx = cx; kx < KS; kx += 2) {
            const int tap = ky * KS + kx;
            uint4 bh[4], bl[4];
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {  // B fragment: output channel r, input channels ks * 16 + h * 8 .. + 7
                bh[ks] = wp[((size_t)(0 * TAPS + tap) * KP8 + ks * 2 + h) * NP + r];
                bl[ks] = wp[((size_t)(1 * TAPS + tap) * KP8 + ks * 2 + h) * NP + r];
            }
            f16v acc;
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {  // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(al[ks]), as_bf8(bh[ks]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah[ks]), as_bf8(bl[ks]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah[ks]), as_bf8(bh[ks]), acc, 0, 0, 0);
            }
            const int slot = (ky >> 1) * 4 + (kx >> 1);
            float* dst = prod + (size_t)row0 * ROWF + slot * CO + r;  // acc[i]: cell row 8 * (i / 4) + 4 * h + i % 4, channel r
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int rr = 8 * (i >> 2) + 4 * h + (i & 3);
                if (row0 + rr < end) dst[(size_t)rr * ROWF] = acc[i];
            }
        }
    }
}

// ---- 3. every output pixel gathers the tap products of its window ------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_gather_kernel(const unsigned* __restrict__ bitmap, int words, const int* __restrict__ cell_pos,
                                                          const float* __restrict__ prod, const float* __restrict__ bias, int hi, int wi,
                                                          int ho, int wo, int relu, float* __restrict__ out, float* __restrict__ stats_partial) {
    __shared__ float sv[32][CO + 1];
    const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;           // 8 lanes x 4 channels per pixel, 32 pixels per block
    const long pix = (long)blockIdx.x * 32 + pl;                     // (sample, oy, ox) flattened; wo is a multiple of 32
    const int ox = (int)(pix % wo), oy = (int)((pix / wo) % ho), b = (int)(pix / ((long)wo * ho));
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int ix0 = 2 * ox - PAD;
    for (int ky = 0; ky < KS; ky++) {
        const int iy = 2 * oy - PAD + ky;
        if (iy < 0 || iy >= hi) continue;
        const long row = (long)b * hi + iy;
        // the 7 window bits of this row: columns ix0 .. ix0 + 6 (two bitmap words at most)
        unsigned bits = 0;
        {
            const int lo = ix0 < 0 ? 0 : ix0, hi_c = ix0 + 6 >= wi ? wi - 1 : ix0 + 6;
            const int w0 = lo >> 5, w1 = hi_c >> 5;
            const unsigned long long two = (unsigned long long)bitmap[row * words + w0] |
                                           (w1 != w0 ? (unsigned long long)bitmap[row * words + w1] << 32 : 0ull);
            const int sh = ix0 - (w0 << 5);  // may be negative at the left border
            bits = sh >= 0 ? (unsigned)((two >> sh) & 0x7full) : (unsigned)((two << (-sh)) & 0x7full);
            if (ix0 + 6 >= wi) bits &= (1u << (wi - ix0)) - 1u;
        }
        while (bits) {
            const int kx = __ffs(bits) - 1;
            bits &= bits - 1;
            const int pos = cell_pos[row * wi + ix0 + kx];
            if (pos < 0) continue;
            const float4 v = *reinterpret_cast<const float4*>(prod + (size_t)pos * ROWF + ((ky >> 1) * 4 + (kx >> 1)) * CO + 4 * q);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    *reinterpret_cast<float4*>(out + pix * CO + 4 * q) = acc;
    if (!stats_partial) return;
    sv[pl][4 * q] = acc.x; sv[pl][4 * q + 1] = acc.y; sv[pl][4 * q + 2] = acc.z; sv[pl][4 * q + 3] = acc.w;
    __syncthreads();
    if (threadIdx.x < 2 * CO) {  // per-block sums and sums of squares per channel, pixels in order
        const int c = threadIdx.x & (CO - 1), sq = threadIdx.x >= CO;
        float s = 0.f;
        for (int p = 0; p < 32; p++) s += sq ? sv[p][c] * sv[p][c] : sv[p][c];
        stats_partial[((size_t)blockIdx.x * 2 + sq) * CO + c] = s;
    }
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct Layout {
    size_t cnt4, off4, seg, bitmap, cells, cell_pos, prod, total;
    int n_rows, words, cap;
};

inline bool layout(int batch, int hi, int wi, int max_cells_per_sample, Layout* l) {
    if (batch < 1 || hi < 2 || wi < 64 || (hi & 1) || (wi & 63) || max_cells_per_sample < 1) return false;
    if ((long)batch * hi * wi > (1L << 30)) return false;
    l->n_rows = batch * hi;
    l->words = wi / 32;
    const long cap = (long)batch * max_cells_per_sample + 4 * 128;
    if (cap > (1L << 24)) return false;
    l->cap = (int)(cap / 128 * 128);
    size_t o = 0;
    l->cnt4 = o; o += align256((size_t)4 * l->n_rows * 4);
    l->off4 = o; o += align256((size_t)4 * l->n_rows * 4);
    l->seg = o; o += 256;
    l->bitmap = o; o += align256((size_t)l->n_rows * l->words * 4);
    l->cells = o; o += align256((size_t)l->cap * 4);
    l->cell_pos = o; o += align256((size_t)batch * hi * wi * 4);
    l->prod = o; o += align256((size_t)l->cap * ROWF * 4);
    l->total = o;
    return true;
}

}  // namespace

extern "C" {

size_t liso_sparse_stem_workspace_bytes(int batch, int hi, int wi, int max_cells_per_sample) {
    Layout l;
    return layout(batch, hi, wi, max_cells_per_sample, &l) ? l.total : 0;
}

int liso_sparse_stem_forward_f32(const float* x, long x_pix_stride, const float* occupancy, const void* w_packed, const float* bias,
                                 int batch, int hi, int wi, int max_cells_per_sample, int relu, float* y, float* stats_partial,
                                 int* overflow, void* workspace, size_t workspace_bytes, void* stream) {
    Layout l;
    if (!x || !occupancy || !w_packed || !y || !workspace) return LISO_EINVAL;
    if (x_pix_stride < CI || (x_pix_stride & 3) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed | (uintptr_t)workspace) & 15))
        return LISO_EINVAL;
    if (!layout(batch, hi, wi, max_cells_per_sample, &l)) return LISO_EINVAL;
    if (workspace_bytes < l.total) return LISO_EWORKSPACE;
    const int ho = hi / 2, wo = wi / 2;  // (hi + 2 * 3 - 7) / 2 + 1 for even hi
    if (wo % 32) return LISO_EINVAL;
    char* ws = (char*)workspace;
    int* cnt4 = (int*)(ws + l.cnt4);
    int* off4 = (int*)(ws + l.off4);
    int* seg = (int*)(ws + l.seg);
    unsigned* bitmap = (unsigned*)(ws + l.bitmap);
    int* cells = (int*)(ws + l.cells);
    int* cell_pos = (int*)(ws + l.cell_pos);
    float* prod = (float*)(ws + l.prod);
    hipStream_t st = (hipStream_t)stream;
    cells_rows_kernel<<<l.n_rows, 64, 0, st>>>(occupancy, wi, l.words, l.n_rows, cnt4, bitmap);
    cells_scan_kernel<<<1, 1024, 0, st>>>(cnt4, l.n_rows, l.cap, off4, seg, overflow);
    cells_fill_kernel<<<l.n_rows, 64, 0, st>>>(occupancy, wi, l.n_rows, off4, seg, cells, cell_pos);
    stem_taps_kernel<<<l.cap / 128, 256, 0, st>>>(x, x_pix_stride, cells, seg, (const uint4*)w_packed, prod);
    const long pixels = (long)batch * ho * wo;
    stem_gather_kernel<<<(unsigned)(pixels / 32), 256, 0, st>>>(bitmap, l.words, cell_pos, prod, bias, hi, wi, ho, wo, relu, y, stats_partial);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // extern "C"
