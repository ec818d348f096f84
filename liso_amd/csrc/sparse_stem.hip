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
#include <stdlib.h>

#include "../../include/liso_conv.h"
#include "../../include/liso_iou3d.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int CI = 64;               // input channels (the pillar feature width)
constexpr int NP = 64;               // packed panels: [plane][tap][K / 8][NP][8] bf16 (liso_conv_pack_weights; N padded to 64)
template <int K> struct Geo {        // k x k kernel, stride 2, padding k / 2 (odd): a cell is reached by the taps of its parity class
    static constexpr int P = K / 2, SX = (K + 1) / 2, SLOTS = SX * SX, TAPS = K * K;  // product slot of a tap: (ky >> 1) * SX + (kx >> 1)
};

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
    __shared__ int part[4][256];
    __shared__ int cls_total[4];
    const int cls = threadIdx.x >> 8, t = threadIdx.x & 255;  // 256 threads per class: the four scans run side by side
    const int* c = cnt4 + (size_t)cls * n_rows;
    const int per = (n_rows + 255) / 256;
    const int b0 = t * per;
    int s = 0;
    for (int i0 = 0; i0 < per; i0 += 8) {  // (8 independent loads in flight: one dependent chain per row otherwise)
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (i0 + u < per && b0 + i0 + u < n_rows) ? c[b0 + i0 + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
    }
    part[cls][t] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan of the 256 partial sums of every class
        const int v = t >= o ? part[cls][t - o] : 0;
        __syncthreads();
        part[cls][t] += v;
        __syncthreads();
    }
    int run = t > 0 ? part[cls][t - 1] : 0;
    for (int i0 = 0; i0 < per; i0 += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (i0 + u < per && b0 + i0 + u < n_rows) ? c[b0 + i0 + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (i0 + u < per && b0 + i0 + u < n_rows) {
                off4[(size_t)cls * n_rows + b0 + i0 + u] = run;  // relative to the class base (added by the consumers)
                run += v[u];
            }
    }
    if (t == 255) cls_total[cls] = part[cls][255];
    __syncthreads();
    if (threadIdx.x == 0) {
        int base = 0, over = 0;
        for (int q = 0; q < 4; q++) {
            int end = base + cls_total[q];
            if (end > cap) { end = cap > base ? cap : base; over = 1; }
            seg[q] = base;
            seg[4 + q] = end;
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
// fp32 -> (hi, lo) bf16 fragments of 8 consecutive values
__device__ __forceinline__ void split8(const float* __restrict__ src, uint4& hi, uint4& lo) {
    const float4 v0 = *reinterpret_cast<const float4*>(src);
    const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
    const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned hi2[4], lo2[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
        hi2[e] = pack_bf16(h0, h1);
        lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
    }
    hi = make_uint4(hi2[0], hi2[1], hi2[2], hi2[3]);
    lo = make_uint4(lo2[0], lo2[1], lo2[2], lo2[3]);
}

__device__ __forceinline__ int class_of_block(const int* __restrict__ seg, int pos0) {
    int cls = -1;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (pos0 >= seg[q] && pos0 < seg[4 + q]) cls = q;
    return cls;
}

// BF16: bf16 tensors, one MFMA per product; otherwise fp32 tensors in F32X3 arithmetic (hi * hi + hi * lo + lo * hi, small terms first)
template <int K, int CO, bool BF16>
__global__ __launch_bounds__(256) void stem_taps_kernel(const void* __restrict__ xv, long xps, const int* __restrict__ cells,
                                                        const int* __restrict__ seg, const uint4* __restrict__ wp,
                                                        float* __restrict__ prod) {
    using G = Geo<K>;
    constexpr int ROWF = G::SLOTS * CO, NT = CO / 32, KP8 = CI / 8;
    // 32 cells per block; the taps of the cells' class are dealt to the four waves (a block of 128 cells with every wave walking all
    // 16 taps of the 7x7 kernel took 46 us for 8 sweeps: one dependent load -> MFMA -> store chain per tap)
    const int row0 = blockIdx.x * 32;
    const int cls = class_of_block(seg, row0);
    if (cls < 0) return;
    const int end = seg[4 + cls];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    // A fragments: the features of cell row0 + r, channels ks * 16 + h * 8 .. + 7
    uint4 ah[4], al[4];
    {
        const int p = row0 + r;
        const size_t cell = (size_t)cells[p < end ? p : row0];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if constexpr (BF16) {
                ah[ks] = *reinterpret_cast<const uint4*>((const unsigned short*)xv + cell * xps + ks * 16 + h * 8);
                al[ks] = ah[ks];
            } else {
                split8((const float*)xv + cell * xps + ks * 16 + h * 8, ah[ks], al[ks]);
            }
        }
    }
    const int cy = cls >> 1, cx = cls & 1;
    int ti = 0;
    for (int ky = cy; ky < K; ky += 2) {
        for (int kx = cx; kx < K; kx += 2, ti++) {
            if ((ti & 3) != wave) continue;
            const int tap = ky * K + kx;
            const int slot = (ky >> 1) * G::SX + (kx >> 1);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                uint4 bh[4], bl[4];
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {  // B fragment: output channel nt * 32 + r, input channels ks * 16 + h * 8 .. + 7
                    bh[ks] = wp[((size_t)(0 * G::TAPS + tap) * KP8 + ks * 2 + h) * NP + nt * 32 + r];
                    if constexpr (!BF16) bl[ks] = wp[((size_t)(1 * G::TAPS + tap) * KP8 + ks * 2 + h) * NP + nt * 32 + r];
                }
                f16v acc;
#pragma unroll
                for (int i = 0; i < 16; i++) acc[i] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    if constexpr (!BF16) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(al[ks]), as_bf8(bh[ks]), acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah[ks]), as_bf8(bl[ks]), acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah[ks]), as_bf8(bh[ks]), acc, 0, 0, 0);
                }
                float* dst = prod + (size_t)row0 * ROWF + slot * CO + nt * 32 + r;  // acc[i]: cell row 8 * (i / 4) + 4 * h + i % 4
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int rr = 8 * (i >> 2) + 4 * h + (i & 3);
                    if (row0 + rr < end) dst[(size_t)rr * ROWF] = acc[i];
                }
            }
        }
    }
}

// ---- 3. every output pixel gathers the tap products of its window ------------------------------------------------------------------
template <int K, int CO, bool BF16>
__global__ __launch_bounds__(256) void stem_gather_kernel(const unsigned* __restrict__ bitmap, int words, const int* __restrict__ cell_pos,
                                                          const float* __restrict__ prod, const float* __restrict__ bias, int hi, int wi,
                                                          int ho, int wo, int relu, int groups, void* __restrict__ outv,
                                                          float* __restrict__ stats_partial, const float* __restrict__ stats_shift) {
    using G = Geo<K>;
    // 16 channels per lane (the window bookkeeping is per pixel: with 4 channels per lane 16 lanes repeated it and the kernel was
    // bound by exactly that, 42 us for 262k pixels), LPP lanes per pixel, PPB pixels per pass
    constexpr int CPL = 16, ROWF = G::SLOTS * CO, LPP = CO / CPL, PPB = 256 / LPP;
    __shared__ float sv[2][PPB / 4][CO + 1];
    const int q = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    float bias_v[CPL], sh_v[CPL], s1[CPL], s2[CPL];
#pragma unroll
    for (int e = 0; e < CPL; e++) {
        bias_v[e] = bias ? bias[CPL * q + e] : 0.f;
        sh_v[e] = (stats_partial && stats_shift) ? stats_shift[CPL * q + e] : 0.f;
        s1[e] = 0.f;
        s2[e] = 0.f;
    }
    // a block walks `groups` passes of PPB consecutive pixels (the pixels of a sample are a multiple of PPB * groups): one statistics
    // row per block.  (Also measured: one 64-bit window mask per output pixel, set by OR atomics in cells_fill_kernel, instead of the
    // row bitmaps -- the gather went 74 -> 64 us on 8 sweeps, the fill 5 -> 25 us: no gain; two contributions in flight per row: 68 us;
    // the pass's output staged through LDS and stored 16 B per lane back to back: 89 us.)  (Fetching the window bits of all passes and kernel rows up front -- independent loads -- was measured: 43.6
    // instead of 35.5 us, the registers cost more occupancy than the shorter chain gains.)
    for (int g = 0; g < groups; g++) {
        const unsigned pixu = (blockIdx.x * (unsigned)groups + (unsigned)g) * PPB + pl;  // (sample, oy, ox) flattened (< 2^30: layout())
        const unsigned rowu = pixu / (unsigned)wo;                                         // (32-bit divisions)
        const int ox = (int)(pixu - rowu * (unsigned)wo), b = (int)(rowu / (unsigned)ho), oy = (int)(rowu - (unsigned)b * (unsigned)ho);
        const long pix = (long)pixu;
        float acc[CPL];
#pragma unroll
        for (int e = 0; e < CPL; e++) acc[e] = bias_v[e];
        const int ix0 = 2 * ox - G::P;
        for (int ky = 0; ky < K; ky++) {
            const int iy = 2 * oy - G::P + ky;
            if (iy < 0 || iy >= hi) continue;
            const long row = (long)b * hi + iy;
            // the K window bits of this row: columns ix0 .. ix0 + K - 1 (two bitmap words at most)
            unsigned bits = 0;
            {
                const int lo = ix0 < 0 ? 0 : ix0, hi_c = ix0 + K - 1 >= wi ? wi - 1 : ix0 + K - 1;
                const int w0 = lo >> 5, w1 = hi_c >> 5;
                const unsigned long long two = (unsigned long long)bitmap[row * words + w0] |
                                               (w1 != w0 ? (unsigned long long)bitmap[row * words + w1] << 32 : 0ull);
                const int sh = ix0 - (w0 << 5);  // negative at the left border
                bits = sh >= 0 ? (unsigned)((two >> sh) & ((1u << K) - 1u)) : (unsigned)((two << (-sh)) & ((1u << K) - 1u));
                if (ix0 + K - 1 >= wi) bits &= (1u << (wi - ix0)) - 1u;
            }
            while (bits) {
                const int kx = __ffs(bits) - 1;
                bits &= bits - 1;
                const int pos = cell_pos[row * wi + ix0 + kx];
                if (pos < 0) continue;
                const float4* v = reinterpret_cast<const float4*>(prod + (size_t)pos * ROWF + ((ky >> 1) * G::SX + (kx >> 1)) * CO + CPL * q);
#pragma unroll
                for (int e4 = 0; e4 < CPL / 4; e4++) {
                    const float4 t = v[e4];
                    acc[4 * e4] += t.x; acc[4 * e4 + 1] += t.y; acc[4 * e4 + 2] += t.z; acc[4 * e4 + 3] += t.w;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < CPL; e++) {
            if (relu) acc[e] = fmaxf(acc[e], 0.f);
            if constexpr (BF16) acc[e] = round_bf16(acc[e]);  // (statistics of the stored, rounded values, like the dense epilogue)
        }
        if constexpr (BF16) {
            uint4* dst = reinterpret_cast<uint4*>((unsigned short*)outv + pix * CO + CPL * q);
            dst[0] = make_uint4(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]), pack_bf16(acc[4], acc[5]), pack_bf16(acc[6], acc[7]));
            dst[1] = make_uint4(pack_bf16(acc[8], acc[9]), pack_bf16(acc[10], acc[11]), pack_bf16(acc[12], acc[13]), pack_bf16(acc[14], acc[15]));
        } else {
            float4* dst = reinterpret_cast<float4*>((float*)outv + pix * CO + CPL * q);
#pragma unroll
            for (int e4 = 0; e4 < CPL / 4; e4++) dst[e4] = make_float4(acc[4 * e4], acc[4 * e4 + 1], acc[4 * e4 + 2], acc[4 * e4 + 3]);
        }
        if (stats_partial) {
#pragma unroll
            for (int e = 0; e < CPL; e++) {
                const float dd = acc[e] - sh_v[e];
                s1[e] += dd;
                s2[e] = fmaf(dd, dd, s2[e]);
            }
        }
    }
    if (!stats_partial) return;
    // block sums in a fixed order: four pixel lanes at a time through DPP-free shuffles would need LPP-specific code -- two LDS rounds
    // instead (pixel lanes 4 j .. 4 j + 3 are added by their first lane, then PPB / 4 rows by the channel threads)
#pragma unroll
    for (int e = 0; e < CPL; e++) {
        float a = s1[e], c = s2[e];
        a += __shfl_down(a, LPP); c += __shfl_down(c, LPP);
        a += __shfl_down(a, 2 * LPP); c += __shfl_down(c, 2 * LPP);
        if ((pl & 3) == 0) { sv[0][pl >> 2][CPL * q + e] = a; sv[1][pl >> 2][CPL * q + e] = c; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CO) {
        const int c = threadIdx.x % CO, sq = threadIdx.x >= CO;
        float s = 0.f;
        for (int p = 0; p < PPB / 4; p++) s += sv[sq][p][c];
        stats_partial[((size_t)blockIdx.x * 2 + sq) * CO + c] = s;
    }
}

// ---- 4. data gradient at the occupied cells -----------------------------------------------------------------------------------------
// dx[cell][ci] = sum over the taps of the cell's class of dy[(iy + P - ky) / 2][(ix + P - kx) / 2][:] . W[tap][:, ci]: the pillar encoder's
// backward reads the canvas gradient at occupied cells only (the rest of dx is the caller's zero fill).  A = the gathered dy rows of 32
// cells (K dimension = output channels), B = the data-gradient panels of the tap; all taps of the class add into one accumulator.
template <int K, int CO, bool BF16>
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const void* __restrict__ dyv, long gps, const int* __restrict__ cells,
                                                         const int* __restrict__ seg, const uint4* __restrict__ wp, int hi, int wi,
                                                         int ho, int wo, void* __restrict__ dxv, long dxps) {
    using G = Geo<K>;
    constexpr int KSTEPS = CO / 16, KP8 = CO / 8, NT = CI / 32;
    const int pos0 = blockIdx.x * 128;
    const int cls = class_of_block(seg, pos0);
    if (cls < 0) return;
    const int end = seg[4 + cls];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = pos0 + wave * 32;
    if (row0 >= end) return;
    const bool live = row0 + r < end;
    const int cell = cells[live ? row0 + r : row0];
    const int ix = cell % wi, iy = (cell / wi) % hi, b = cell / (wi * hi);
    f16v acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[nt][i] = 0.0f;
    const int cy = cls >> 1, cx = cls & 1;
    for (int ky = cy; ky < K; ky += 2) {
        const int oy = (iy + G::P - ky) / 2;  // (exact: ky has the parity of iy + P)
        for (int kx = cx; kx < K; kx += 2) {
            const int ox = (ix + G::P - kx) / 2;
            const bool ok = live && iy + G::P - ky >= 0 && oy < ho && ix + G::P - kx >= 0 && ox < wo;
            const size_t orow = (((size_t)b * ho + (ok ? oy : 0)) * wo + (ok ? ox : 0)) * gps;
            const int tap = ky * K + kx;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) {
                uint4 ah, al;
                if constexpr (BF16) {
                    ah = ok ? *reinterpret_cast<const uint4*>((const unsigned short*)dyv + orow + ks * 16 + h * 8) : make_uint4(0, 0, 0, 0);
                    al = ah;
                } else {
                    if (ok) split8((const float*)dyv + orow + ks * 16 + h * 8, ah, al);
                    else { ah = make_uint4(0, 0, 0, 0); al = ah; }
                }
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const uint4 bh = wp[((size_t)(0 * G::TAPS + tap) * KP8 + ks * 2 + h) * NP + nt * 32 + r];
                    if constexpr (!BF16) {
                        const uint4 bl = wp[((size_t)(1 * G::TAPS + tap) * KP8 + ks * 2 + h) * NP + nt * 32 + r];
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(al), as_bf8(bh), acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah), as_bf8(bl), acc[nt], 0, 0, 0);
                    }
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah), as_bf8(bh), acc[nt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {  // acc[nt][i]: cell row 8 * (i / 4) + 4 * h + i % 4, input channel nt * 32 + r
        const int rr = 8 * (i >> 2) + 4 * h + (i & 3);
        const int c2 = __shfl(cell, rr);  // (lane rr holds the cell of row rr)
        if (row0 + rr < end) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                if constexpr (BF16) {
                    const __bf16 v = (__bf16)acc[nt][i];
                    ((unsigned short*)dxv)[(size_t)c2 * dxps + nt * 32 + r] = __builtin_bit_cast(unsigned short, v);
                } else {
                    ((float*)dxv)[(size_t)c2 * dxps + nt * 32 + r] = acc[nt][i];
                }
            }
        }
    }
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct Layout {
    size_t cnt4, off4, seg, bitmap, cells, cell_pos, prod, total;
    int n_rows, words, cap;
};

inline bool geometry_ok(int k, int co) { return (k == 3 || k == 7) && (co == 32 || co == 64); }

inline bool layout(int batch, int hi, int wi, int k, int co, int max_cells_per_sample, bool with_products, Layout* l) {
    if (batch < 1 || hi < 2 || wi < 64 || (hi & 1) || (wi & 63) || max_cells_per_sample < 1 || !geometry_ok(k, co)) return false;
    if ((long)batch * hi * wi > (1L << 30)) return false;
    l->n_rows = batch * hi;
    l->words = wi / 32;
    const long cap = (long)batch * max_cells_per_sample + 4 * 128;
    if (cap > (1L << 24)) return false;
    l->cap = (int)(cap / 128 * 128);
    const int sx = (k + 1) / 2;
    size_t o = 0;
    l->cnt4 = o; o += align256((size_t)4 * l->n_rows * 4);
    l->off4 = o; o += align256((size_t)4 * l->n_rows * 4);
    l->seg = o; o += 256;
    l->bitmap = o; o += align256((size_t)l->n_rows * l->words * 4);
    l->cells = o; o += align256((size_t)l->cap * 4);
    l->cell_pos = o; o += align256((size_t)batch * hi * wi * 4);
    l->prod = o;
    if (with_products) o += align256((size_t)l->cap * sx * sx * co * 4);
    l->total = o;
    return true;
}

void cell_lists(const float* occupancy, int wi, const Layout& l, char* ws, int* overflow, hipStream_t st) {
    int* cnt4 = (int*)(ws + l.cnt4);
    int* off4 = (int*)(ws + l.off4);
    int* seg = (int*)(ws + l.seg);
    cells_rows_kernel<<<l.n_rows, 64, 0, st>>>(occupancy, wi, l.words, l.n_rows, cnt4, (unsigned*)(ws + l.bitmap));
    cells_scan_kernel<<<1, 1024, 0, st>>>(cnt4, l.n_rows, l.cap, off4, seg, overflow);
    cells_fill_kernel<<<l.n_rows, 64, 0, st>>>(occupancy, wi, l.n_rows, off4, seg, (int*)(ws + l.cells), (int*)(ws + l.cell_pos));
}

}  // namespace

extern "C" {

int liso_sparse_conv_stat_groups(int hi, int wi, int co) {
    if (hi < 2 || wi < 64 || (co != 32 && co != 64)) return 0;
    const long per_sample = (long)(hi / 2) * (wi / 2);
    const int ppb = 256 / (co / 16);
    int g = 4;
    if (const char* e = getenv("LISO_SPARSE_GROUPS")) g = atoi(e) >= 1 && atoi(e) <= 4 ? atoi(e) : 4;  // experiments
    while (g > 1 && per_sample % (ppb * g)) g >>= 1;
    return per_sample % (ppb * g) ? 0 : g;
}

size_t liso_sparse_conv_workspace_bytes(int batch, int hi, int wi, int k, int co, int max_cells_per_sample, int for_dgrad) {
    Layout l;
    return layout(batch, hi, wi, k, co, max_cells_per_sample, !for_dgrad, &l) ? l.total : 0;
}

int liso_sparse_conv_forward(const void* x, long x_pix_stride, int is_bf16, const float* occupancy, const void* w_packed,
                             const float* bias, int batch, int hi, int wi, int k, int co, int max_cells_per_sample, int relu, void* y,
                             float* stats_partial, const float* stats_shift, int* overflow, void* workspace, size_t workspace_bytes,
                             void* stream) {
    Layout l;
    if (!x || !occupancy || !w_packed || !y || !workspace) return LISO_EINVAL;
    const int vec = is_bf16 ? 8 : 4;
    if (x_pix_stride < CI || (x_pix_stride % vec) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed | (uintptr_t)workspace) & 15))
        return LISO_EINVAL;
    if (!layout(batch, hi, wi, k, co, max_cells_per_sample, true, &l)) return LISO_EINVAL;
    if (workspace_bytes < l.total) return LISO_EWORKSPACE;
    const int ho = hi / 2, wo = wi / 2;  // (hi + 2 * (k / 2) - k) / 2 + 1 for even hi, odd k
    const int ppb = 256 / (co / 16);
    const int groups = liso_sparse_conv_stat_groups(hi, wi, co);
    if (groups <= 0) return LISO_EINVAL;
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    cell_lists(occupancy, wi, l, ws, overflow, st);
    const int* cells = (const int*)(ws + l.cells);
    const int* seg = (const int*)(ws + l.seg);
    const unsigned* bitmap = (const unsigned*)(ws + l.bitmap);
    const int* cell_pos = (const int*)(ws + l.cell_pos);
    float* prod = (float*)(ws + l.prod);
    const unsigned tb = (unsigned)(l.cap / 32), gb = (unsigned)((long)batch * ho * wo / ((long)ppb * groups));
#define LISO_SPARSE_FWD(K, CO, BF)                                                                                                     \
    do {                                                                                                                                \
        stem_taps_kernel<K, CO, BF><<<tb, 256, 0, st>>>(x, x_pix_stride, cells, seg, (const uint4*)w_packed, prod);                     \
        stem_gather_kernel<K, CO, BF><<<gb, 256, 0, st>>>(bitmap, l.words, cell_pos, prod, bias, hi, wi, ho, wo, relu, groups, y, stats_partial, \
                                                         stats_shift);                                                                 \
    } while (0)
    if (k == 7 && co == 32 && !is_bf16) LISO_SPARSE_FWD(7, 32, false);
    else if (k == 3 && co == 64 && is_bf16) LISO_SPARSE_FWD(3, 64, true);
    else if (k == 3 && co == 64 && !is_bf16) LISO_SPARSE_FWD(3, 64, false);
    else return LISO_EINVAL;
#undef LISO_SPARSE_FWD
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

int liso_sparse_conv_dgrad(const void* dy, long dy_pix_stride, int is_bf16, const float* occupancy, const void* w_packed_dgrad, int batch,
                           int hi, int wi, int k, int co, int max_cells_per_sample, void* dx, long dx_pix_stride, int* overflow,
                           void* workspace, size_t workspace_bytes, int reuse_lists, void* stream) {
    Layout l;
    if (!dy || (!occupancy && !reuse_lists) || !w_packed_dgrad || !dx || !workspace) return LISO_EINVAL;
    const int vec = is_bf16 ? 8 : 4;
    if (dy_pix_stride < co || (dy_pix_stride % vec) || dx_pix_stride < CI || (((uintptr_t)dy | (uintptr_t)w_packed_dgrad | (uintptr_t)workspace) & 15))
        return LISO_EINVAL;
    if (!layout(batch, hi, wi, k, co, max_cells_per_sample, false, &l)) return LISO_EINVAL;
    if (workspace_bytes < l.total) return LISO_EWORKSPACE;
    const int ho = hi / 2, wo = wi / 2;
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (!reuse_lists) cell_lists(occupancy, wi, l, ws, overflow, st);  // (else: the workspace of the forward call on the same canvas)
    const int* cells = (const int*)(ws + l.cells);
    const int* seg = (const int*)(ws + l.seg);
    const unsigned tb = (unsigned)(l.cap / 128);
#define LISO_SPARSE_DG(K, CO, BF) \
    stem_dgrad_kernel<K, CO, BF><<<tb, 256, 0, st>>>(dy, dy_pix_stride, cells, seg, (const uint4*)w_packed_dgrad, hi, wi, ho, wo, dx, dx_pix_stride)
    if (k == 7 && co == 32 && !is_bf16) LISO_SPARSE_DG(7, 32, false);
    else if (k == 3 && co == 64 && is_bf16) LISO_SPARSE_DG(3, 64, true);
    else if (k == 3 && co == 64 && !is_bf16) LISO_SPARSE_DG(3, 64, false);
    else return LISO_EINVAL;
#undef LISO_SPARSE_DG
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

size_t liso_sparse_stem_workspace_bytes(int batch, int hi, int wi, int max_cells_per_sample) {
    return liso_sparse_conv_workspace_bytes(batch, hi, wi, 7, 32, max_cells_per_sample, 0);
}

int liso_sparse_stem_forward_f32(const float* x, long x_pix_stride, const float* occupancy, const void* w_packed, const float* bias,
                                 int batch, int hi, int wi, int max_cells_per_sample, int relu, float* y, float* stats_partial,
                                 int* overflow, void* workspace, size_t workspace_bytes, void* stream) {
    return liso_sparse_conv_forward(x, x_pix_stride, 0, occupancy, w_packed, bias, batch, hi, wi, 7, 32, max_cells_per_sample, relu, y,
                                    stats_partial, nullptr, overflow, workspace, workspace_bytes, stream);
}

}  // extern "C"
