// NHWC implicit-GEMM convolutions on the gfx950 matrix cores: forward / data gradient (one kernel) -- C ABI and the
// reference lines it replaces: include/liso_conv.h.  The weight gradient lives in conv_wgrad.hip.
//
// Work decomposition.  A 256-thread block (4 wavefronts) owns TH x 32 virtual pixels (TH = 4 or 8) x BNT output channels
// (32 or 64) of one sample.  The reduction runs over channel slabs (CS = 16/32/64 input channels) x taps:
//   * per slab the input halo tile ((TH-1)*is + kh_span) x (31*is + kw_span) pixels x CS channels is staged ONCE into LDS
//     (16-B global loads -> registers -> [BatchNorm-apply + ReLU of the producing layer | fp32 -> bf16 hi/lo split] ->
//     LDS), so every input element is read from HBM/L2 once per block and used by all taps from LDS;
//   * the tap's weight panel [CS/8][BNT][8] bf16 (8 KB) is double-buffered in LDS: the panel of tap t+1 is loaded into
//     registers before and written to LDS after the MFMAs of tap t (one barrier per tap);
//   * wave w computes pixels [w*M/4, (w+1)*M/4) x all BNT channels: per 16-channel k-step MI A-fragments (pixels,
//     ds_read_b128: lane = pixel row, 8 consecutive channels) + NJ B-fragments (ds_read_b128: lane = output channel, 8
//     consecutive input channels of the pre-packed panel) feed MI x NJ v_mfma_f32_32x32x16_bf16.
// LDS images: a pixel occupies CS*2 + 16 bytes (odd multiple of 16): the 16 lanes of a ds_read_b128 group read 16
// different pixels at the same channel offset and land on 16 different 16-B bank groups (conflict-free; 2-way for the
// stride-2 layers).  65-80 KB of LDS and <128 VGPRs per block -> 2 blocks per CU: one block's staging overlaps the other's
// MFMAs.  The blockIdx -> tile map hands every XCD a contiguous range of tiles (its L2 holds the halo rows it re-reads).
//
// F32X3: fp32 tensors.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi); a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
// (3 MFMAs, fp32 accumulate, small terms first).  Two LDS planes (hi, lo) for tile and panel.
//
// Epilogue: C/D layout of the 32x32 MFMA = (column = lane & 31 = output channel, row = pixel).  fp32 outputs: one dword per
// lane and register, 128-B runs per pixel.  bf16 outputs: neighbouring lanes exchange one value per register pair through
// DPP (quad_perm 1,0,3,2), so every lane stores 2 adjacent channels of one pixel (4 B; 64-B runs per pixel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liso_conv.h"
#include "../../include/liso_iou3d.h"
#include "conv_plan.h"
#include "per_device.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int xcd_remap(int bid, int total) {
    // blocks b and b + 8 share an XCD (round-robin dispatch): give every XCD a contiguous chunk of the logical ids
    const int q = total >> 3, r = total & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float round_bf16(float v) { return (float)(__bf16)v; }

__device__ __forceinline__ bf8 as_bf8(const uint4& v) { return __builtin_bit_cast(bf8, v); }

// ---- epilogue shared by the forward kernels ---------------------------------------------------------------------------------------
// acc[i][j]: the 32 x 32 MFMA tiles of wave `wave` (tile row wave * MI + i of the block's TH = 4 MI rows, output channels n0 + 32 j ...).
// `active` = this thread holds accumulators (waves 0-3 of the block, split-K group 0); every thread of the block must call (barriers).
template <int MI, int NJ, bool OUT_F32, bool WAVE_STATS = false>
__device__ __forceinline__ void conv_epilogue(const liso_conv_desc& d, const FwdArgs& a, f16v (&acc)[MI][NJ], int cls, int b, int tx, int ty,
                                              int wave, int r, int h, bool active, int n0, int stats_row, int tid_all,
                                              unsigned char* smem) {
    constexpr int BNT = 32 * NJ;
    constexpr int TH = 4 * MI;
    const int tid = tid_all & (kThreads - 1);
    const int grp = active ? 0 : 1;
    // ---- epilogue ------------------------------------------------------------------------------------------------------------
    // Register e of a 32 x 32 tile is pixel column (e & 3) + 8 (e >> 2) + 4 h of ONE tile row (TW = 32): the row part of the
    // output offset and the row validity are per (wave, i); the column part is a compile-time multiple of the pixel stride.
    const bool want_stats = a.stats != nullptr;
    float s1[NJ], s2[NJ];
    const int ooy = d.class_ooy[cls], oox = d.class_oox[cls];
    const int col_stride = d.osx * d.y_pix_stride;  // elements between horizontally adjacent virtual pixels
    unsigned colmask = 0;                           // bit e: the pixel column of register e exists
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int vx = tx * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (vx < d.wv && vx * d.osx + oox < d.wo) colmask |= 1u << e;
    }
    const int col0 = (tx * 32 + 4 * h) * d.osx + oox;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        s1[j] = 0.0f;
        s2[j] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < MI; i++) {
        const int vy = ty * TH + wave * MI + i;
        const int oy = vy * d.osy + ooy;
        const unsigned rowmask = (grp == 0 && vy < d.hv && oy < d.ho) ? colmask : 0u;
        if constexpr (WAVE_STATS) {
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                s1[j] = 0.0f;
                s2[j] = 0.0f;
            }
        }
        const long row_base = (((long)b * d.ho + oy) * d.wo + col0) * d.y_pix_stride + d.y_ch_off;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int n = n0 + j * 32 + r;
            const bool n_ok = n < d.co;
            const float bias_v = (a.bias && n_ok) ? a.bias[n] : 0.0f;
            const float shift_v = (a.stats_shift && n_ok) ? a.stats_shift[n] : 0.0f;
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                float val = acc[i][j][e] + bias_v;
                if (d.out_relu) val = fmaxf(val, 0.0f);
                if constexpr (!OUT_F32) val = round_bf16(val);
                v[e] = val;
                if constexpr (!WAVE_STATS) {
                    if (want_stats && ((rowmask >> e) & 1u)) {
                        const float dd = val - shift_v;
                        s1[j] += dd;
                        s2[j] = fmaf(dd, dd, s2[j]);
                    }
                }
            }
            if constexpr (WAVE_STATS) {
                if (want_stats) {  // pairwise sums over the lane's 16 pixels (error grows with log n, not n)
                    float q1[16], q2[16];
#pragma unroll
                    for (int e = 0; e < 16; e++) {
                        const float dd = ((rowmask >> e) & 1u) ? v[e] - shift_v : 0.0f;
                        q1[e] = dd;
                        q2[e] = dd * dd;
                    }
#pragma unroll
                    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                        for (int e = 0; e < w; e++) {
                            q1[e] += q1[e + w];
                            q2[e] += q2[e + w];
                        }
                    s1[j] = q1[0];
                    s2[j] = q2[0];
                }
            }
            if constexpr (OUT_F32) {
                float* yg = (float*)a.y + row_base + n;
#pragma unroll
                for (int e = 0; e < 16; e++)
                    if (((rowmask >> e) & 1u) && n_ok) yg[((e & 3) + 8 * (e >> 2)) * col_stride] = v[e];
            } else {
                unsigned short* yg = (unsigned short*)a.y + row_base;
                const bool odd = r & 1;
                const int n_even = n & ~1;
                if ((d.y_pix_stride | d.y_ch_off) & 1) {  // odd pixel stride: channel pairs are not 4-B aligned, 2-B stores
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        if (((rowmask >> e) & 1u) && n_ok)
                            yg[((e & 3) + 8 * (e >> 2)) * col_stride + n] = (unsigned short)(pack_bf16(v[e], 0.0f) & 0xffffu);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const float send = odd ? v[e] : v[e + 1];
                        const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, true));
                        const float c_lo = odd ? recv : v[e];      // channel n_even
                        const float c_hi = odd ? v[e + 1] : recv;  // channel n_even + 1
                        const int ee = odd ? e + 1 : e;            // the register (pixel) this lane stores: compile-time per parity
                        const int koff = odd ? ((e + 1) & 3) + 8 * ((e + 1) >> 2) : (e & 3) + 8 * (e >> 2);
                        if ((rowmask >> ee) & 1u) {
                            unsigned short* dst = yg + koff * col_stride + n_even;
                            if (n_even + 1 < d.co)
                                *reinterpret_cast<unsigned*>(dst) = pack_bf16(c_lo, c_hi);
                            else if (n_even < d.co)
                                *dst = (unsigned short)(pack_bf16(c_lo, 0.0f) & 0xffffu);
                        }
                    }
                }
            }
        }
        if constexpr (WAVE_STATS) {
            if (want_stats) {  // sums of tile row (wave * MI + i) -> LDS [TH][BNT][2]
                float* red = reinterpret_cast<float*>(smem);
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const float t1 = s1[j] + __shfl_xor(s1[j], 32);
                    const float t2 = s2[j] + __shfl_xor(s2[j], 32);
                    if (h == 0) {
                        red[((wave * MI + i) * BNT + j * 32 + r) * 2 + 0] = t1;
                        red[((wave * MI + i) * BNT + j * 32 + r) * 2 + 1] = t2;
                    }
                }
            }
        }
    }
    if constexpr (WAVE_STATS) {
        // (conv_roles_kernel: the sums of every 32-pixel tile row went to LDS inside the row loop; roles_flush_stats adds them behind
        // the block's next barrier)
    } else if (want_stats) {
        float* red = reinterpret_cast<float*>(smem);  // [4 waves][BNT][2]; the main loop ended with a barrier
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float t1 = s1[j] + __shfl_xor(s1[j], 32);
            const float t2 = s2[j] + __shfl_xor(s2[j], 32);
            if (h == 0 && grp == 0) {
                red[(wave * BNT + j * 32 + r) * 2 + 0] = t1;
                red[(wave * BNT + j * 32 + r) * 2 + 1] = t2;
            }
        }
        __syncthreads();
        if (tid_all < BNT && n0 + tid < a.co_pad) {  // (a 96-filter panel may reach beyond the padded filter count: no statistics columns there)
            float q1 = 0.0f, q2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                q1 += red[(w * BNT + tid) * 2 + 0];
                q2 += red[(w * BNT + tid) * 2 + 1];
            }
            a.stats[((long)stats_row * 2 + 0) * a.co_pad + n0 + tid] = q1;
            a.stats[((long)stats_row * 2 + 1) * a.co_pad + n0 + tid] = q2;
        }
    }
}

// SK = 2 (small maps, one sample: 100-200 blocks of one wave per SIMD, where the slab loop is bound by the latency of its own
// loads): the block has two groups of 4 waves, each with its own tile + panel buffers, that take alternate channel slabs (twice
// the loads in flight per CU, half the slab iterations); group 1 hands its accumulators over through LDS before the epilogue.
template <int MODE, int MI, int NJ, bool OUT_F32, int CS, int SK = 1>
__global__ __launch_bounds__(kThreads * SK, SK == 1 ? 2 : 1) void conv_igemm_kernel(const liso_conv_desc d, const FwdArgs a) {
    constexpr int BNT = 32 * NJ;
    constexpr int TH = 4 * MI;
    constexpr bool X3 = MODE == LISO_CONV_F32X3;
    constexpr bool F32 = MODE == LISO_CONV_F32;  // exact fp32: v_mfma_f32_32x32x2_f32 on fp32 tiles / panels (one plane)
    constexpr bool FIN = X3 || F32;              // fp32 tensors in HBM
    constexpr int PLANES = X3 ? 2 : 1;
    constexpr int PS = (F32 ? CS * 4 : CS * 2) + 16;  // LDS bytes per pixel and plane
    constexpr int KS = F32 ? CS / 8 : CS / 16;  // fragment reads per tap: one ds_read_b128 = 8 bf16 | 4 fp32 per lane, both lane halves
    constexpr int KG = F32 ? 4 : 8;             // channels per 16-B group of a weight panel
    constexpr int K8 = CS / KG;                 // 16-B channel groups per slab
    constexpr int PSZ = K8 * BNT;              // 16-B chunks of one weight panel (one tap, one plane)
    constexpr int WTAP = PSZ * 16;             // bytes of one panel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid_all = threadIdx.x, grp = SK == 1 ? 0 : tid_all >> 8;
    const int tid = tid_all & (kThreads - 1), wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    // sparse inputs: occupied tiles cluster (the sensor sits in the middle of the BEV map).  Blocks b and b + 8 share an XCD: dealt in
    // plain order every XCD would own one tile COLUMN, with contiguous ranges one group of rows -- either way a few XCDs get all the
    // occupied tiles.  So the tiles are dealt round-robin with the column rotated by the row (every XCD sees every column).
    int t = a.occ ? (int)blockIdx.x : xcd_remap(blockIdx.x, a.total);
    const int nt = t % a.n_nt;
    t /= a.n_nt;
    int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    t /= a.tiles_y;
    const int b = t % d.batch;
    const int cls = t / d.batch;
    if (a.occ) tx = (tx + ty + b) % a.tiles_x;
    const int stats_row = ((cls * d.batch + b) * a.tiles_y + ty) * a.tiles_x + tx;
    const int n0 = nt * BNT;

    const int tb = d.class_tap_begin[cls], te = d.class_tap_begin[cls + 1];
    const int dy0 = a.cls_dy0[cls], dx0 = a.cls_dx0[cls], in_h = a.cls_inh[cls], in_w = a.cls_inw[cls];
    const int npix = in_h * in_w;
    const float inv_w = 1.0f / (float)in_w;
    const int iy0 = ty * TH * d.isy + dy0, ix0 = tx * 32 * d.isx + dx0;

    // LDS: [tap tables 512 B][input tile, PLANES planes][weight panels of one stage: [tap][plane][K8][BNT][8]]
    int* s_toff = reinterpret_cast<int*>(smem);          // byte offset of the tap inside the input tile
    int* s_tapw = s_toff + 64;                           // tap index inside the packed weights
    unsigned char* xs = smem + 512 + (SK == 1 ? 0 : grp * a.group_bytes);
    unsigned char* wsb = xs + a.x_plane_bytes * PLANES;
    const int G = a.g_taps;
    if (tid_all < te - tb) {
        s_toff[tid_all] = ((d.tap_dy[tb + tid_all] - dy0) * in_w + (d.tap_dx[tb + tid_all] - dx0)) * PS;
        s_tapw[tid_all] = d.tap_w[tb + tid_all];
    }
    // Sparse input (the pillar canvas in front of the encoders' first convolution: a few percent of the BEV cells hold points): a
    // block whose whole halo tile is unoccupied multiplies exact zeros -- its accumulators stay +0 and it goes straight to the
    // epilogue (bias / ReLU / statistics), bit for bit what the full computation gives, without reading the tile or the weights.
    bool tile_empty = false;
    if (a.occ) {
        int any = 0;
        const float* oc = a.occ + (long)b * d.hi * d.wi;
        for (int pix = tid_all; pix < npix; pix += kThreads * SK) {
            const int ly = pix / in_w, lx = pix - ly * in_w;
            const int iy = iy0 + ly, ix = ix0 + lx;
            if ((unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi) any |= oc[iy * d.wi + ix] != 0.0f;
        }
        // block-wide OR through the (not yet used) front of the tile buffer: __syncthreads_or would add static LDS on top of the
        // 160 KB dynamic allocation
        int* flag = reinterpret_cast<int*>(smem + 512);  // (group 0's buffer: one flag for the whole block)
        if (tid_all == 0) *flag = 0;
        __syncthreads();
        if (any) *flag = 1;
        __syncthreads();
        tile_empty = *flag == 0;
        __syncthreads();
    }

    int a_off[MI];
#pragma unroll
    for (int i = 0; i < MI; i++) {
        const int m = wave * (32 * MI) + i * 32 + r;
        a_off[i] = (((m >> 5) * d.isy) * in_w + (m & 31) * d.isx) * PS + h * 16;
    }
    int b_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) b_off[j] = (h * BNT + j * 32 + r) * 16;

    f16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    const int n_taps = te - tb;
    const int kgroups_total = a.ci_pad / KG;
    const long x_img = (long)b * d.hi * d.wi;
    const unsigned short* wg = (const unsigned short*)a.w;
    const long plane_elems = (long)d.w_taps * kgroups_total * a.co_pad * 8;

    // ---- staging pieces: global -> registers (load_*) and registers -> LDS (store_*), so that the loads of slab k+1 can be
    // in flight while the MFMAs of slab k run ---------------------------------------------------------------------------------
    constexpr int CPP = FIN ? CS / 4 : CS / 8;  // 16-B chunks per pixel on the global side (4 fp32 | 8 bf16)
    constexpr int CHN = FIN ? 4 : 8;            // channels per chunk
    constexpr int pstep = kThreads / CPP;
    constexpr int XB = 12, WB = 10;        // 16-B loads per thread and batch
    const int cx = tid % CPP, p0 = tid / CPP;
    const int step_y = pstep / in_w, step_x = pstep - step_y * in_w;  // (uniform)
    const unsigned char* xbase = (const unsigned char*)a.x + x_img * d.x_pix_stride * (FIN ? 4 : 2);
    const bool pro = a.in_scale != nullptr;
    const int aff_off = b * d.in_affine_batch_stride;  // per-sample prologue vectors (InstanceNorm) or 0

    auto load_x = [&](int c0, int pix_begin, uint4 (&v)[XB], unsigned& okmask) {
        const int ch = c0 + cx * CHN;
        const bool ch_ok = ch < d.ci;
        const int pfirst = pix_begin + p0;
        int ly = (int)(((float)pfirst + 0.5f) * inv_w);
        int lx = pfirst - ly * in_w;
        okmask = 0u;
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int pix = pfirst + u * pstep;
            const int iy = iy0 + ly, ix = ix0 + lx;
            const bool ok = pix < npix && ch_ok && (unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi;
            okmask |= ok ? (1u << u) : 0u;
            const int off = ok ? (iy * d.wi + ix) * d.x_pix_stride + ch : 0;  // (a sample has < 2^31 elements)
            v[u] = *reinterpret_cast<const uint4*>(xbase + (long)off * (FIN ? 4 : 2));
            lx += step_x;
            ly += step_y;
            if (lx >= in_w) {
                lx -= in_w;
                ly++;
            }
        }
    };
    auto store_x = [&](int c0, int pix_begin, const uint4 (&v)[XB], unsigned okmask) {
        const int ch = c0 + cx * CHN;
        const bool ch_ok = ch < d.ci;
        float sc[CHN], sh[CHN];
        if (pro) {
#pragma unroll
            for (int e = 0; e < CHN; e++) {
                sc[e] = ch_ok ? a.in_scale[aff_off + ch + e] : 0.0f;
                sh[e] = ch_ok ? a.in_shift[aff_off + ch + e] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int pix = pix_begin + p0 + u * pstep;
            if (pix >= npix) continue;
            const bool ok = (okmask >> u) & 1u;
            if constexpr (!FIN) {
                uint4 o = v[u];
                if (pro) {
                    unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float f0 = fmaf(bf16_lo(w[e]), sc[2 * e], sh[2 * e]);
                        float f1 = fmaf(bf16_hi(w[e]), sc[2 * e + 1], sh[2 * e + 1]);
                        if (d.in_relu) {
                            f0 = fmaxf(f0, 0.0f);
                            f1 = fmaxf(f1, 0.0f);
                        }
                        w[e] = pack_bf16(f0, f1);
                    }
                    o = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (!ok) o = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(xs + pix * PS + cx * 16) = o;
            } else {
                float f[4] = {__uint_as_float(v[u].x), __uint_as_float(v[u].y), __uint_as_float(v[u].z), __uint_as_float(v[u].w)};
                unsigned hi2[2], lo2[2];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (pro) {
                        f[e] = fmaf(f[e], sc[e], sh[e]);
                        if (d.in_relu) f[e] = fmaxf(f[e], 0.0f);
                    }
                    if (!ok) f[e] = 0.0f;
                }
                if constexpr (F32) {
                    *reinterpret_cast<float4*>(xs + pix * PS + cx * 16) = make_float4(f[0], f[1], f[2], f[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
                        hi2[e] = pack_bf16(h0, h1);
                        lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
                    }
                    *reinterpret_cast<uint2*>(xs + pix * PS + cx * 8) = make_uint2(hi2[0], hi2[1]);
                    *reinterpret_cast<uint2*>(xs + a.x_plane_bytes + pix * PS + cx * 8) = make_uint2(lo2[0], lo2[1]);
                }
            }
        }
    };
    auto load_w = [&](int c0, int s0, int chunks, int q_begin, uint4 (&v)[WB]) {
#pragma unroll
        for (int u = 0; u < WB; u++) {
            const int q = q_begin + tid + u * kThreads;
            const int panel = q / PSZ, inner = q % PSZ;  // (compile-time powers of two)
            const int g = panel / PLANES, plane = panel % PLANES;
            const int c8 = inner / BNT, n = inner % BNT;
            const int kg = c0 / KG + c8;
            const bool okq = q < chunks && kg < kgroups_total;
            const int tw = s_tapw[s0 + (okq ? g : 0)];
            const int off = okq ? (plane * (int)plane_elems + ((tw * kgroups_total + kg) * a.co_pad + n0 + n) * 8) : 0;
            v[u] = *reinterpret_cast<const uint4*>(wg + off);
            if (!okq) v[u] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_w = [&](int chunks, int q_begin, const uint4 (&v)[WB]) {
#pragma unroll
        for (int u = 0; u < WB; u++) {
            const int q = q_begin + tid + u * kThreads;
            if (q < chunks) *reinterpret_cast<uint4*>(wsb + q * 16) = v[u];
        }
    };
    auto mfma_taps = [&](int s0, int g_cur) {
        for (int g = 0; g < g_cur; g++) {
            const int toff = s_toff[s0 + g];
            const unsigned char* wt = wsb + g * (PLANES * WTAP);
            uint4 af[KS][MI], bfr[KS][NJ];
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
#pragma unroll
                for (int i = 0; i < MI; i++) af[kk][i] = *reinterpret_cast<const uint4*>(xs + a_off[i] + toff + kk * 32);
#pragma unroll
                for (int j = 0; j < NJ; j++) bfr[kk][j] = *reinterpret_cast<const uint4*>(wt + b_off[j] + kk * (2 * BNT * 16));
            }
            if constexpr (F32) {
                // lane (r, h) holds channels kk*8 + 4h + e of its pixel / output channel: MFMA e pairs channel kk*8 + e (k = 0) with
                // kk*8 + 4 + e (k = 1) on BOTH operands.  Exact fp32 products, one rounding per accumulation step.
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
#pragma unroll
                    for (int i = 0; i < MI; i++)
#pragma unroll
                        for (int j = 0; j < NJ; j++) {
                            const uint4 av = af[kk][i], bv = bfr[kk][j];
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.x), __uint_as_float(bv.x), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.y), __uint_as_float(bv.y), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.z), __uint_as_float(bv.z), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av.w), __uint_as_float(bv.w), acc[i][j], 0, 0, 0);
                        }
            } else if constexpr (X3) {
                uint4 al[KS][MI], bl[KS][NJ];
#pragma unroll
                for (int kk = 0; kk < KS; kk++) {
#pragma unroll
                    for (int i = 0; i < MI; i++)
                        al[kk][i] = *reinterpret_cast<const uint4*>(xs + a.x_plane_bytes + a_off[i] + toff + kk * 32);
#pragma unroll
                    for (int j = 0; j < NJ; j++) bl[kk][j] = *reinterpret_cast<const uint4*>(wt + WTAP + b_off[j] + kk * (2 * BNT * 16));
                }
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
#pragma unroll
                    for (int i = 0; i < MI; i++)
#pragma unroll
                        for (int j = 0; j < NJ; j++) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(al[kk][i]), as_bf8(bfr[kk][j]), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(af[kk][i]), as_bf8(bl[kk][j]), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(af[kk][i]), as_bf8(bfr[kk][j]), acc[i][j], 0, 0, 0);
                        }
            } else {
#pragma unroll
                for (int kk = 0; kk < KS; kk++)
#pragma unroll
                    for (int i = 0; i < MI; i++)
#pragma unroll
                        for (int j = 0; j < NJ; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(af[kk][i]), as_bf8(bfr[kk][j]), acc[i][j], 0, 0, 0);
            }
        }
    };

    if (tile_empty) {
        // nothing to accumulate
    } else if constexpr (SK > 1) {
        // (always pipelined) the groups take alternate slabs; both run the same number of barriers
        uint4 xv[XB], wv[WB];
        unsigned xok = 0u;
        const int chunks = n_taps * PLANES * PSZ;
        const int nslab = (d.ci + CS - 1) / CS, iters = (nslab + SK - 1) / SK;
        __syncthreads();  // tap tables
        int c0 = grp * CS;
        bool have = c0 < d.ci;
        if (have) {
            load_x(c0, 0, xv, xok);
            load_w(c0, 0, chunks, 0, wv);
        }
        for (int it = 0; it < iters; it++) {
            if (it > 0) __syncthreads();  // every read of the previous slab's tile / panels is done
            if (have) {
                store_x(c0, 0, xv, xok);
                store_w(chunks, 0, wv);
            }
            __syncthreads();
            const int cn = c0 + SK * CS;
            const bool have_n = cn < d.ci;
            if (have_n) {
                load_x(cn, 0, xv, xok);
                load_w(cn, 0, chunks, 0, wv);
            }
            if (have) mfma_taps(0, n_taps);
            c0 = cn;
            have = have_n;
        }
        __syncthreads();
        // group 1 -> group 0: accumulators through LDS ([register][thread]: conflict-free), then one epilogue
        float* xfer = reinterpret_cast<float*>(smem + 512);
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++) xfer[((i * NJ + j) * 16 + e) * kThreads + tid] = acc[i][j][e];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[i][j][e] += xfer[((i * NJ + j) * 16 + e) * kThreads + tid];
        }
    } else if (a.pipelined) {
        // Software pipeline over STAGES (a stage = the panels of G taps of one channel slab, one register batch; the halo tile of a
        // slab, one register batch, travels with the slab's first stage): the global loads of stage t + 1 are in flight while the MFMAs
        // of stage t run.  Round 4 pipelined only layers whose whole slab (all taps) fits one register batch -- every 3x3 layer with
        // 64-channel panels ran load -> store -> barrier -> multiply strictly in sequence, 3 exposed load latencies per slab (measured
        // round 5: 8 us per slab on the ConvGRU layers, 1.8 us of it MFMA).
        uint4 xv[XB], wv[WB];
        unsigned xok;
        __syncthreads();  // tap tables
        load_x(0, 0, xv, xok);
        load_w(0, 0, min(G, n_taps) * PLANES * PSZ, 0, wv);
        bool first = true;
        for (int c0 = 0; c0 < d.ci; c0 += CS) {
            for (int s0 = 0; s0 < n_taps; s0 += G) {
                const int g_cur = min(G, n_taps - s0);
                if (!first) __syncthreads();  // every read of the previous stage's panels (and, at s0 == 0, of the previous slab's tile) is done
                first = false;
                if (s0 == 0) store_x(c0, 0, xv, xok);
                store_w(g_cur * PLANES * PSZ, 0, wv);
                __syncthreads();
                int ns0 = s0 + G, nc0 = c0;
                if (ns0 >= n_taps) {
                    ns0 = 0;
                    nc0 = c0 + CS;
                }
                if (nc0 < d.ci) {
                    if (ns0 == 0) load_x(nc0, 0, xv, xok);
                    load_w(nc0, ns0, min(G, n_taps - ns0) * PLANES * PSZ, 0, wv);
                }
                mfma_taps(s0, g_cur);
            }
        }
    } else {
        for (int c0 = 0; c0 < d.ci; c0 += CS) {
            __syncthreads();  // every read of the previous slab's tile / panels is done (and the tap tables are written)
            for (int pb = 0; pb < npix; pb += XB * pstep) {
                uint4 xv[XB];
                unsigned xok;
                load_x(c0, pb, xv, xok);
                store_x(c0, pb, xv, xok);
            }
            for (int s0 = 0; s0 < n_taps; s0 += G) {
                const int g_cur = min(G, n_taps - s0);
                if (s0 > 0) __syncthreads();  // the previous stage's panels have been read
                const int chunks = g_cur * PLANES * PSZ;
                for (int qb = 0; qb < chunks; qb += WB * kThreads) {
                    uint4 wv[WB];
                    load_w(c0, s0, chunks, qb, wv);
                    store_w(chunks, qb, wv);
                }
                __syncthreads();
                mfma_taps(s0, g_cur);
            }
        }
    }
    __syncthreads();

    conv_epilogue<MI, NJ, OUT_F32>(d, a, acc, cls, b, tx, ty, wave, r, h, grp == 0, n0, stats_row, tid_all, smem);
}

// BatchNorm / InstanceNorm partial sums of conv_roles_kernel: one statistics row per 4 tile rows x 32 pixels = sum of the four 32-pixel row
// sums the epilogue left in LDS, ((r0 + r1) + (r2 + r3)) -- a fixed tree over fixed pixel sets, whatever the tile shape, the panel width
// or the tile -> block map (results do not change with the batch size the plan was made for).  Rows: ((b * tiles_y + ty) * MI + g) * tiles_x + tx.
template <int MI, int NJ>
__device__ __forceinline__ void roles_flush_stats(const FwdArgs& a, int b, int ty, int tx, int n0, int tid_all, const unsigned char* smem) {
    constexpr int BNT = 32 * NJ;
    // (a 96-channel panel may reach beyond the padded filter count -- co = 64: co_pad = 64 -- where the statistics rows have no columns:
    // found by tests/test_gpu_conv_roles.py with a forced tile shape; the planner's own choices never had such a panel write statistics)
    if (tid_all >= BNT || n0 + tid_all >= a.co_pad) return;
    const float* red = reinterpret_cast<const float*>(smem);
#pragma unroll
    for (int g = 0; g < MI; g++) {
        const long row = ((long)(b * a.tiles_y + ty) * MI + g) * a.tiles_x + tx;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const float p0 = red[((4 * g + 0) * BNT + tid_all) * 2 + q], p1 = red[((4 * g + 1) * BNT + tid_all) * 2 + q];
            const float p2 = red[((4 * g + 2) * BNT + tid_all) * 2 + q], p3 = red[((4 * g + 3) * BNT + tid_all) * 2 + q];
            a.stats[(row * 2 + q) * a.co_pad + n0 + tid_all] = (p0 + p1) + (p2 + p3);
        }
    }
}

// Epilogue of conv_roles_kernel with WIDE stores: the 32 x 32 MFMA tile layout (lane = output channel, register = pixel) gives one
// dword (fp32) or one channel pair (bf16) per lane and store instruction -- 16 / 8 store instructions per tile, and the s_memtime stamps
// of round 5 showed the epilogue at 10-15 k cycles per tile, more than the multiplications of a shallow layer.  Here every wave turns its
// tile round through a private 2-KB LDS patch, 16 pixels at a time ([pixel][32 channels], written as it lies in the registers, read back
// as 16 bytes per lane): 4 (fp32) or 2 (bf16) 16-byte store instructions per tile, whole 128-B / 64-B runs per pixel.  No barrier: a wave
// only reads what it wrote.  Bias, ReLU, rounding and the statistics sums (pairwise per lane, one LDS row per 32-pixel tile row: see
// roles_flush_stats) as in conv_epilogue.  `wide` false (channel counts / strides that do not allow 16-byte stores): conv_epilogue.
template <int MI, int NJ, bool OUT_F32>
__device__ __forceinline__ void roles_epilogue(const liso_conv_desc& d, const FwdArgs& a, f16v (&acc)[MI][NJ], int b, int tx, int ty, int wave,
                                               int lane, int n0, unsigned char* smem, unsigned char* patch) {
    constexpr int BNT = 32 * NJ;
    constexpr int TH = 4 * MI;
    const int r = lane & 31, h = lane >> 5;
    const bool want_stats = a.stats != nullptr;
    const int ooy = d.class_ooy[0], oox = d.class_oox[0];
    float* red = reinterpret_cast<float*>(smem);
    constexpr int ROWB = OUT_F32 ? 128 : 64;   // bytes of one pixel's 32 channels in the patch
    constexpr int CPR = ROWB / 16;             // 16-byte chunks per pixel
    constexpr int PPI = 64 / CPR;              // pixels one store instruction covers (8 | 16)
    constexpr int ES = OUT_F32 ? 4 : 2;
#pragma unroll
    for (int i = 0; i < MI; i++) {
        const int vy = ty * TH + wave * MI + i;
        const int oy = vy + ooy;
        const bool row_ok = vy < d.hv && oy < d.ho;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int n = n0 + j * 32 + r;
            const bool n_ok = n < d.co;
            const float bias_v = (a.bias && n_ok) ? a.bias[n] : 0.0f;
            const float shift_v = (a.stats_shift && n_ok) ? a.stats_shift[n] : 0.0f;
            float v[16], q1[16], q2[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                float val = acc[i][j][e] + bias_v;
                if (d.out_relu) val = fmaxf(val, 0.0f);
                if constexpr (!OUT_F32) val = round_bf16(val);
                v[e] = val;
                const int vx = tx * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const float dd = (row_ok && vx < d.wv && vx + oox < d.wo) ? val - shift_v : 0.0f;
                q1[e] = dd;
                q2[e] = dd * dd;
            }
            if (want_stats) {
#pragma unroll
                for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                    for (int e = 0; e < w; e++) {
                        q1[e] += q1[e + w];
                        q2[e] += q2[e + w];
                    }
                const float t1 = q1[0] + __shfl_xor(q1[0], 32), t2 = q2[0] + __shfl_xor(q2[0], 32);
                if (h == 0) {
                    red[((wave * MI + i) * BNT + j * 32 + r) * 2 + 0] = t1;
                    red[((wave * MI + i) * BNT + j * 32 + r) * 2 + 1] = t2;
                }
            }
            // two halves of 16 pixels: registers e with (e >> 3) == half hold pixel columns 16 half + (e & 3) + 8 ((e >> 2) & 1) + 4 h
#pragma unroll
            for (int half = 0; half < 2; half++) {
                if constexpr (OUT_F32) {
#pragma unroll
                    for (int e8 = 0; e8 < 8; e8++) {
                        const int px = (e8 & 3) + 8 * (e8 >> 2) + 4 * h;
                        *reinterpret_cast<float*>(patch + px * ROWB + r * 4) = v[half * 8 + e8];
                    }
                } else {
                    // neighbouring lanes exchange one value per register pair (DPP quad_perm 1,0,3,2): even lanes keep the pixel of
                    // register e, odd lanes that of e + 1, both as the channel pair (r & ~1, r | 1)
                    const bool odd = r & 1;
#pragma unroll
                    for (int e8 = 0; e8 < 8; e8 += 2) {
                        const int e = half * 8 + e8;
                        const float send = odd ? v[e] : v[e + 1];
                        const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, true));
                        const float c_lo = odd ? recv : v[e], c_hi = odd ? v[e + 1] : recv;
                        const int ee = odd ? e8 + 1 : e8;
                        const int px = (ee & 3) + 8 * (ee >> 2) + 4 * h;
                        *reinterpret_cast<unsigned*>(patch + px * ROWB + (r >> 1) * 4) = pack_bf16(c_lo, c_hi);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 16 / PPI; q++) {
                    const int px = q * PPI + lane / CPR, ck = lane % CPR;
                    const uint4 val = *reinterpret_cast<const uint4*>(patch + px * ROWB + ck * 16);
                    const int vx = tx * 32 + half * 16 + px;
                    const int ch = n0 + j * 32 + ck * (16 / ES);
                    if (row_ok && vx < d.wv && vx + oox < d.wo && ch < d.co) {
                        const long off = (((long)b * d.ho + oy) * d.wo + vx + oox) * d.y_pix_stride + d.y_ch_off + ch;
                        *reinterpret_cast<uint4*>((unsigned char*)a.y + off * ES) = val;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the patch is rewritten by the next half)
            }
        }
    }
}

// ---- loader waves + MFMA waves, double-buffered LDS, persistent blocks: 3x3 / stride 1 / one tap class -----------------------------------
// Round-5 PMC of conv_igemm_kernel on the ConvGRU layer (304 -> 192 at 4 x 64 x 64, F32X3): waves parked at s_waitcnt / s_barrier
// 47 % of their cycles, 7 150 non-MFMA vector instructions per wave beside 1 080 MFMAs (address arithmetic and the hi / lo split are
// re-done by the waves that multiply, in phases that alternate with the MFMA phases), MFMA pipe 18 % busy -- and prefetching the
// global loads one stage ahead changed nothing (94 vs 98 us): the kernel is bound by its own phase structure, not by load latency.
// Here the two jobs run side by side in one block of 8 waves (one block per CU, at most one block per CU in the grid):
//   * waves 4-7 LOAD: global -> registers (two stages in flight) -> [prologue, hi / lo split] -> LDS buffer (k + 1) & 1: the input halo
//     tile of one channel slab and the weight panels of ALL taps of that slab.  The loads of stage k + 2 and the stores of stage k + 1
//     are interleaved chunk by chunk; per-thread addresses are computed once per tile, a slab only adds a stride;
//   * waves 0-3 MULTIPLY on buffer k & 1: taps unrolled with compile-time LDS offsets (the loaders store the panels in window order),
//     fragments of tap t + 1 are requested before the MFMAs of tap t issue; nothing but ds_read_b128 and v_mfma in the loop;
//   * one raw s_barrier per slab (LDS counters drained, vector-memory loads left in flight);
//   * a block walks tiles blockIdx, blockIdx + gridDim, ...: the stages of all its tiles form ONE stream, so the loaders stage the next
//     tile's first slabs while the MFMA waves run the epilogue of the previous one (bias / ReLU / stores / BatchNorm sums: one statistics
//     row per (tile, wave), no LDS and no barrier in the epilogue).
// CS = 16 channels per slab for fp32 tensors (F32X3: two bf16 planes), 32 for bf16 tensors: (tile + 9 panels) x 2 buffers = 113 KB at
// 64-channel panels, 150 KB at 96.  LDS images as in conv_igemm_kernel (pixel stride CS * 2 + 16 B: conflict-free ds_read_b128).
// Measured (in-kernel s_memtime stamps, `make STAMPS=1` + scripts/roles_stamps.py): the loaders move ~17-19 B / clk / CU from L2 whatever
// the instruction mix -- a 4-row tile with 96-channel panels (73 KB per slab) is bound by them (4 300 cycles per slab against 2 600 of
// MFMA), 8-row tiles with 64-channel panels are balanced.
template <int MODE, int MI, int NJ, bool OUT_F32, int NTAPS, bool PRO>
__global__ __launch_bounds__(512, 1) void conv_roles_kernel(const liso_conv_desc d, const FwdArgs a) {
    constexpr bool X3 = MODE == LISO_CONV_F32X3;
    constexpr int PLANES = X3 ? 2 : 1;
    constexpr int CS = X3 ? 16 : 32;
    constexpr int BNT = 32 * NJ;
    constexpr int TH = 4 * MI;
    constexpr int PS = CS * 2 + 16;
    constexpr int KS = CS / 16;
    constexpr int K8 = CS / 8;
    constexpr int PSZ = K8 * BNT;        // 16-B chunks of one panel (one tap, one plane)
    constexpr int WTAP = PSZ * 16;       // bytes of one panel
    constexpr int KW = NTAPS == 9 ? 3 : 1;
    constexpr int IN_W = 32 + KW - 1, IN_H = TH + KW - 1, NPIX = IN_H * IN_W;
    constexpr int XPLANE = NPIX * PS;    // bytes of one plane of the tile
    constexpr int WSTAGE = NTAPS * PLANES * WTAP;
    constexpr int BUF = (XPLANE * PLANES + WSTAGE + 15) / 16 * 16;
    constexpr int CPP = 4;               // 16-B chunks per tile pixel on the global side (fp32: 16 ch / 4, bf16: 32 ch / 8)
    constexpr int CHN = X3 ? 4 : 8;      // channels per chunk
    constexpr int PSTEP = 256 / CPP;
    constexpr int XB = (NPIX * CPP + 255) / 256;
    constexpr int WCH = NTAPS * PLANES * PSZ;  // weight chunks per stage
    constexpr int WB = (WCH + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid_all = threadIdx.x, wave_all = tid_all >> 6, lane = tid_all & 63, r = lane & 31, h = lane >> 5;
    const bool loader = wave_all >= 4;
    const int wave = wave_all & 3;
    const int nslab = (d.ci + CS - 1) / CS;
    const int dy0 = a.cls_dy0[0], dx0 = a.cls_dx0[0];
    unsigned char* base = smem + 4096;  // (front: the epilogue's row sums [TH][BNT][2] fp32)
    // tiles of this block: logical ids j * G + chunk(blockIdx) -- within a round of G blocks every XCD (blocks b, b + 8, ... share one)
    // gets a contiguous range of tiles (its L2 holds the halo rows and the panels they share)
    const int G = gridDim.x;
    const int bid = blockIdx.x;
    const int chunked = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : xcd_remap(bid, G);
    const int my_tiles = chunked < a.total ? (a.total - 1 - chunked) / G + 1 : 0;
    const int S = my_tiles * nslab;  // stages of this block
    struct Tile {
        int b, ty, tx, n0, row;
    };
    auto tile_of = [&](int j) {
        int t = j * G + chunked;
        Tile T;
        const int nt = t % a.n_nt;
        t /= a.n_nt;
        T.tx = t % a.tiles_x;
        t /= a.tiles_x;
        T.ty = t % a.tiles_y;
        T.b = t / a.tiles_y;
        T.n0 = nt * BNT;
        T.row = (T.b * a.tiles_y + T.ty) * a.tiles_x + T.tx;
        return T;
    };
    auto barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    if (S == 0) return;

    if (loader) {
        const int tid = tid_all - 256;
        const int cx = tid % CPP, p0 = tid / CPP;
        constexpr bool pro = PRO;  // (a template parameter: a run-time branch around the prologue's loads makes hipcc's s_waitcnt insertion
                                   // drain ALL loads in front of the first LDS store of a stage -- measured: loads and stores then add up)
        const unsigned short* wg = (const unsigned short*)a.w;
        const int kgroups_total = a.ci_pad >> 3;
        const long plane_elems = (long)d.w_taps * kgroups_total * a.co_pad * 8;
        const int w_slab = K8 * a.co_pad * 8;  // elements between consecutive slabs of one tap
        // tile-independent per-thread constants: LDS byte of every tile chunk; weight chunk q = (window position g, plane, 8-channel
        // group c8, output channel n) in LDS order -> element offset at n0 = 0 and the chunk's output channel
        int x_lds[XB], w_off[WB], w_n[WB];
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int pix = p0 + u * PSTEP;
            x_lds[u] = pix < NPIX ? pix * PS + cx * (X3 ? 8 : 16) : -1;
        }
#pragma unroll
        for (int u = 0; u < WB; u++) {
            const int q = tid + u * 256;
            const int panel = q / PSZ, inner = q % PSZ;
            const int g = panel / PLANES, plane = panel % PLANES;
            const int c8 = inner / BNT, n = inner % BNT;
            const bool ok = q < WCH;
            const int tw = (int)((a.roles_tapw >> (4 * (ok ? g : 0))) & 15ull);  // (a packed word: indexing a kernel-argument array by a register goes through scratch)
            w_off[u] = ok ? (int)(plane * plane_elems) + ((tw * kgroups_total + c8) * a.co_pad + n) * 8 : 0;
            w_n[u] = ok ? n : (1 << 30);
        }
        // the ISSUE stream's position and the per-tile constants of its tile
        int is_tile = 0, is_slab = 0;
        int x_off[XB];
        unsigned xmask = 0u, wmask = 0u;
        int n0e = 0, aff_off = 0;
        const unsigned char* xbase = nullptr;
        auto enter_tile = [&](int j) {
            const Tile T = tile_of(j);
            const int iy0 = T.ty * TH + dy0, ix0 = T.tx * 32 + dx0;
            xbase = (const unsigned char*)a.x + (long)T.b * d.hi * d.wi * d.x_pix_stride * (X3 ? 4 : 2);
            aff_off = T.b * d.in_affine_batch_stride;
            n0e = T.n0 * 8;
            xmask = 0u;
            wmask = 0u;
#pragma unroll
            for (int u = 0; u < XB; u++) {
                const int pix = p0 + u * PSTEP;
                const int ly = pix / IN_W, lx = pix - ly * IN_W;
                const int iy = iy0 + ly, ix = ix0 + lx;
                const bool ok = pix < NPIX && (unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi;
                xmask |= ok ? (1u << u) : 0u;
                x_off[u] = ok ? (iy * d.wi + ix) * d.x_pix_stride + cx * CHN : 0;
            }
#pragma unroll
            for (int u = 0; u < WB; u++) wmask |= (T.n0 + w_n[u] < a.co_pad) ? (1u << u) : 0u;
        };
        enter_tile(0);

        struct Regs {
            uint4 x[XB];
            uint4 w[WB];
            float sc[CHN], sh[CHN];
            unsigned xmask, wmask;  // of the stage these registers hold (the issue stream may already be in the next tile)
            bool ch_ok;
        };
        // Loads are nothing but loads (no select on a loaded value: that would wait for it right there) from addresses that always exist;
        // what must be zero is zeroed when it is stored.  Beyond the last stage the stream stays where it is (redundant loads): the loop
        // body has no branch around a load, so hipcc's s_waitcnt insertion counts the younger stage's loads instead of draining everything.
        struct Ctx {
            int cadd, kadd, cc;
        };
        auto begin_stage = [&](Regs& R) {
            Ctx c;
            const int c0 = is_slab * CS;
            const int chx = c0 + cx * CHN;
            R.ch_ok = chx < d.ci;
            R.xmask = xmask;
            R.wmask = wmask;
            c.cadd = R.ch_ok ? c0 : -1;       // (a channel chunk beyond ci: the tensor's first bytes, zeroed when stored)
            c.kadd = is_slab * w_slab + n0e;  // (bf16 layers take this kernel only when ci is a multiple of the 32-channel slab)
            c.cc = R.ch_ok ? aff_off + chx : 0;
            return c;
        };
        auto advance = [&]() {  // (VALU only: no memory operation inside the branch)
            if (is_slab + 1 < nslab) {
                is_slab++;
            } else if (is_tile + 1 < my_tiles) {
                is_tile++;
                is_slab = 0;
                enter_tile(is_tile);
            }
        };
        auto load_w1 = [&](const Ctx& c, Regs& R, int u) {
            R.w[u] = *reinterpret_cast<const uint4*>(wg + w_off[u] + (((wmask >> u) & 1u) ? c.kadd : 0));
        };
        auto load_x1 = [&](const Ctx& c, Regs& R, int u) {
            const bool ok = ((xmask >> u) & 1u) && c.cadd >= 0;
            R.x[u] = *reinterpret_cast<const uint4*>(xbase + (long)(ok ? x_off[u] + c.cadd : 0) * (X3 ? 4 : 2));
        };
        auto load_aff = [&](const Ctx& c, Regs& R) {
            if constexpr (pro) {
#pragma unroll
                for (int e = 0; e < CHN; e++) {
                    R.sc[e] = a.in_scale[c.cc + e];
                    R.sh[e] = a.in_shift[c.cc + e];
                }
            }
        };
        auto store_w1 = [&](const Regs& R, unsigned char* buf, int u) {
            const int q = tid + u * 256;
            if (q < WCH) *reinterpret_cast<uint4*>(buf + XPLANE * PLANES + q * 16) = ((R.wmask >> u) & 1u) ? R.w[u] : make_uint4(0u, 0u, 0u, 0u);
        };
        auto store_x1 = [&](const Regs& R, unsigned char* xs, int u) {
            if (x_lds[u] < 0) return;
            const bool ok = ((R.xmask >> u) & 1u) && R.ch_ok;
            if constexpr (!X3) {
                uint4 o = R.x[u];
                if constexpr (pro) {
                    unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float f0 = fmaf(bf16_lo(w[e]), R.sc[2 * e], R.sh[2 * e]);
                        float f1 = fmaf(bf16_hi(w[e]), R.sc[2 * e + 1], R.sh[2 * e + 1]);
                        if (d.in_relu) {
                            f0 = fmaxf(f0, 0.0f);
                            f1 = fmaxf(f1, 0.0f);
                        }
                        w[e] = pack_bf16(f0, f1);
                    }
                    o = make_uint4(w[0], w[1], w[2], w[3]);
                }
                if (!ok) o = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(xs + x_lds[u]) = o;
            } else {
                float f[4] = {__uint_as_float(R.x[u].x), __uint_as_float(R.x[u].y), __uint_as_float(R.x[u].z), __uint_as_float(R.x[u].w)};
                unsigned hi2[2], lo2[2];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if constexpr (pro) {
                        f[e] = fmaf(f[e], R.sc[e], R.sh[e]);
                        if (d.in_relu) f[e] = fmaxf(f[e], 0.0f);
                    }
                    if (!ok) f[e] = 0.0f;
                }
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
                    hi2[e] = pack_bf16(h0, h1);
                    lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
                }
                *reinterpret_cast<uint2*>(xs + x_lds[u]) = make_uint2(hi2[0], hi2[1]);
                *reinterpret_cast<uint2*>(xs + XPLANE + x_lds[u]) = make_uint2(lo2[0], lo2[1]);
            }
        };
        auto issue = [&](Regs& R) {  // the loads of the stream's current stage -> R; the stream moves on
            const Ctx c = begin_stage(R);
#pragma unroll
            for (int u = 0; u < XB; u++) load_x1(c, R, u);
            load_aff(c, R);
#pragma unroll
            for (int u = 0; u < WB; u++) load_w1(c, R, u);
            advance();
        };
        auto store = [&](const Regs& R, unsigned char* buf) {
#pragma unroll
            for (int u = 0; u < XB; u++) store_x1(R, buf, u);
#pragma unroll
            for (int u = 0; u < WB; u++) store_w1(R, buf, u);
        };
        // loads of the stream's current stage -> RN interleaved with: registers RC -> buf (do_store = false: loads only)
        auto step = [&](Regs& RN, const Regs& RC, unsigned char* buf, bool do_store) {
            const Ctx cn = begin_stage(RN);
#pragma unroll
            for (int u = 0; u < XB; u++) {
                load_x1(cn, RN, u);
                if (u == 0) load_aff(cn, RN);
                if (do_store) store_x1(RC, buf, u);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < WB; u++) {
                load_w1(cn, RN, u);
                if (do_store) store_w1(RC, buf, u);
                __builtin_amdgcn_sched_barrier(0);
            }
            advance();
        };

#ifdef LISO_ROLES_STAMPS
        unsigned long long st_issue = 0, st_store = 0, st_bar = 0, st_t;
#define STAMP_BEGIN st_t = __builtin_amdgcn_s_memtime();
#define STAMP_END(acc) { const unsigned long long st_n = __builtin_amdgcn_s_memtime(); acc += st_n - st_t; st_t = st_n; }
        const unsigned long long st_k0 = __builtin_amdgcn_s_memtime();
#else
#define STAMP_BEGIN
#define STAMP_END(acc)
#endif
        Regs RA, RB;
        issue(RA);  // stage 0
        issue(RB);  // stage 1 (or stage 0 again)
        store(RA, base);
        barrier();  // buffer 0 is ready
#ifdef LISO_ROLES_STAMPS
        const unsigned long long st_k1 = __builtin_amdgcn_s_memtime();
#endif
        // iteration k: the MFMA waves multiply buffer k & 1; here: loads of stage k + 2 -> the register set that was just stored,
        // registers of stage k + 1 -> buffer (k + 1) & 1 (its readers finished before the previous barrier).  S barriers in all.
        int k = 0;
        for (; k + 1 < S; k += 2) {
            STAMP_BEGIN
            step(RA, RB, base + BUF, true);
            STAMP_END(st_issue)
            barrier();
            STAMP_END(st_bar)
            step(RB, RA, base, k + 2 < S);
            STAMP_END(st_store)
            barrier();
            STAMP_END(st_bar)
        }
        if (k < S) barrier();  // (odd stage count: the last multiplication)
        if (a.stats) barrier();  // (the MFMA waves' last statistics rows: roles_flush_stats)
#ifdef LISO_ROLES_STAMPS
        if (a.stats && tid == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(a.stats) + (size_t)blockIdx.x * 16 + 8;
            o[0] = st_k1 - st_k0; o[1] = st_issue; o[2] = st_store; o[3] = st_bar;
        }
#endif
    } else {
        int a_off[MI];
#pragma unroll
        for (int i = 0; i < MI; i++) a_off[i] = ((wave * MI + i) * IN_W + r) * PS + h * 16;
        const int b_off = XPLANE * PLANES + (h * BNT + r) * 16;
        struct Frag {
            uint4 ah[KS][MI], bh[KS][NJ];
            uint4 al[X3 ? KS : 1][X3 ? MI : 1], bl[X3 ? KS : 1][X3 ? NJ : 1];
        };
        f16v acc[MI][NJ];
        auto read = [&](const unsigned char* buf, int tap, Frag& F) {
            const int toff = ((tap / KW) * IN_W + (tap % KW)) * PS;
            const int woff = tap * PLANES * WTAP;
#pragma unroll
            for (int kk = 0; kk < KS; kk++) {
#pragma unroll
                for (int i = 0; i < MI; i++) {
                    F.ah[kk][i] = *reinterpret_cast<const uint4*>(buf + a_off[i] + toff + kk * 32);
                    if constexpr (X3) F.al[kk][i] = *reinterpret_cast<const uint4*>(buf + XPLANE + a_off[i] + toff + kk * 32);
                }
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    F.bh[kk][j] = *reinterpret_cast<const uint4*>(buf + b_off + woff + j * 512 + kk * (2 * BNT * 16));
                    if constexpr (X3) F.bl[kk][j] = *reinterpret_cast<const uint4*>(buf + b_off + woff + WTAP + j * 512 + kk * (2 * BNT * 16));
                }
            }
        };
        auto mul = [&](const Frag& F) {
#pragma unroll
            for (int kk = 0; kk < KS; kk++)
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        if constexpr (X3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(F.al[kk][i]), as_bf8(F.bh[kk][j]), acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(F.ah[kk][i]), as_bf8(F.bl[kk][j]), acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(F.ah[kk][i]), as_bf8(F.bh[kk][j]), acc[i][j], 0, 0, 0);
                    }
        };
#ifdef LISO_ROLES_STAMPS
        unsigned long long st_mul = 0, st_bar = 0, st_epi = 0, st_t;
        const unsigned long long st_k0 = __builtin_amdgcn_s_memtime();
#endif
        barrier();  // buffer 0 is ready
#ifdef LISO_ROLES_STAMPS
        const unsigned long long st_k1 = __builtin_amdgcn_s_memtime();
#endif
        int k = 0;  // stage counter of the block
        for (int j = 0; j < my_tiles; j++) {
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int jj = 0; jj < NJ; jj++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[i][jj][e] = 0.0f;
            __builtin_amdgcn_s_setprio(2);  // the loader waves' vector instructions fill the MFMA shadows, never the other way round
            for (int sl = 0; sl < nslab; sl++, k++) {
                const unsigned char* buf = base + (k & 1) * BUF;
                // fragments of tap t + 1 are requested before the MFMAs of tap t issue (the scheduling barriers keep hipcc from sinking
                // the reads to their first use: it then waits for each group right after asking for it)
                Frag F0, F1;
                STAMP_BEGIN
                read(buf, 0, F0);
#pragma unroll
                for (int tp = 0; tp < NTAPS; tp += 2) {
                    if (tp + 1 < NTAPS) read(buf, tp + 1, F1);
                    __builtin_amdgcn_sched_barrier(0);
                    mul(F0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tp + 1 < NTAPS) {
                        if (tp + 2 < NTAPS) read(buf, tp + 2, F0);
                        __builtin_amdgcn_sched_barrier(0);
                        mul(F1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                STAMP_END(st_mul)
                barrier();
                STAMP_END(st_bar)
#ifndef LISO_ROLES_STAMPS
                if (sl == 0 && j > 0 && a.stats) {  // the previous tile's row sums are complete in LDS (written before this barrier)
                    const Tile P = tile_of(j - 1);
                    roles_flush_stats<MI, NJ>(a, P.b, P.ty, P.tx, P.n0, tid_all, smem);
                }
#endif
            }
            __builtin_amdgcn_s_setprio(0);
            const Tile T = tile_of(j);
#ifdef LISO_ROLES_STAMPS
            FwdArgs a2 = a;
            a2.stats = nullptr;
            conv_epilogue<MI, NJ, OUT_F32, true>(d, a2, acc, 0, T.b, T.tx, T.ty, wave, r, h, true, T.n0, T.row, tid_all, smem);
            STAMP_END(st_epi)
#else
            if (a.wide_out)
                roles_epilogue<MI, NJ, OUT_F32>(d, a, acc, T.b, T.tx, T.ty, wave, lane, T.n0, smem, base + 2 * BUF + wave * 2048);
            else
                conv_epilogue<MI, NJ, OUT_F32, true>(d, a, acc, 0, T.b, T.tx, T.ty, wave, r, h, true, T.n0, T.row, tid_all, smem);
#endif
        }
#ifndef LISO_ROLES_STAMPS
        if (a.stats) {
            barrier();
            const Tile P = tile_of(my_tiles - 1);
            roles_flush_stats<MI, NJ>(a, P.b, P.ty, P.tx, P.n0, tid_all, smem);
        }
#else
        if (a.stats) barrier();
#endif
#ifdef LISO_ROLES_STAMPS
        if (a.stats && tid_all == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(a.stats) + (size_t)blockIdx.x * 16;
            o[0] = st_k1 - st_k0; o[1] = st_mul; o[2] = st_bar; o[3] = __builtin_amdgcn_s_memtime() - st_k0; o[4] = st_epi;
        }
#endif
    }
}

// ---- 1x1 convolutions (any stride) and narrow 3x3 layers, F32X3: fragments straight from global memory ---------------------------------
// Tile geometry, epilogue and statistics rows as conv_igemm_kernel (4 waves x MI rows of 32 pixels x 32 NJ filters per block), but no
// staging: lane (r, h) of a wave reads channels [k0 + 8 h, k0 + 8 h + 8) of ITS pixel r shifted by the tap -- two 16-byte loads, the
// pending BatchNorm / InstanceNorm + ReLU of the producer applied in registers, split into bf16 hi / lo -- and the weight fragments come
// packed in fragment order ([tap][k8][n][8]: 16 bytes per lane, 512 contiguous bytes per lane half).  The steps (tap, 16 channels) form
// one stream; the loads of step t + 1 are in flight under the MFMAs of step t; nothing synchronises before the epilogue.
// One tap: every 1x1 layer.  The step stream walks any tap list, but several taps re-read the input per tap through L1 (32 lines per
// load instruction): measured slower than staging even for one 32-filter panel (plan_1x1, conv_plan.h) -- experiments only.
// Tap classes (transposed convolutions with kernel = stride: every output pixel of class (oy % s, ox % s) reads ONE input pixel through
// ONE tap): a block works on one class -- a 1x1 convolution whose epilogue scatters to that class's output pixels.
// MODE: F32X3 (fp32 tensors, bf16 hi / lo split in registers) or BF16 (bf16 tensors: the lane's 8 channels are one 16-byte load).
template <int MODE, int MI, int NJ, bool OUT_F32, bool PRO>
__global__ __launch_bounds__(kThreads, 2) void conv_1x1_kernel(const liso_conv_desc d, const FwdArgs a) {
    constexpr bool X3 = MODE == LISO_CONV_F32X3;
    constexpr int BNT = 32 * NJ;
    constexpr int TH = 4 * MI;
    constexpr int ES = X3 ? 4 : 2;  // bytes per input element
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 96 * 2 * 4];
    __shared__ int s_dy[LISO_CONV_MAX_TAPS], s_dx[LISO_CONV_MAX_TAPS], s_w[LISO_CONV_MAX_TAPS];
    const int tid_all = threadIdx.x, wave = tid_all >> 6, lane = tid_all & 63, r = lane & 31, h = lane >> 5;
    if (tid_all < d.n_taps) {
        s_dy[tid_all] = d.tap_dy[tid_all];
        s_dx[tid_all] = d.tap_dx[tid_all];
        s_w[tid_all] = d.tap_w[tid_all];
    }
    int t = xcd_remap(blockIdx.x, a.total);
    const int nt = t % a.n_nt;
    t /= a.n_nt;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    t /= a.tiles_y;
    const int b = t % d.batch;
    const int cls = t / d.batch;
    const int stats_row = ((cls * d.batch + b) * a.tiles_y + ty) * a.tiles_x + tx;
    const int n0 = nt * BNT;
    // taps of this block: its class's one tap, or (one class) the whole list
    const int tap0 = d.n_classes > 1 ? cls : 0, tap1 = d.n_classes > 1 ? cls + 1 : d.n_taps;
    const unsigned char* xb = (const unsigned char*)a.x + (long)b * d.hi * d.wi * d.x_pix_stride * ES;
    int by[MI];
    bool okv[MI];
    const int bx = (tx * 32 + r) * d.isx;
#pragma unroll
    for (int i = 0; i < MI; i++) {
        const int vy = ty * TH + wave * MI + i;
        okv[i] = vy < d.hv && tx * 32 + r < d.wv;
        by[i] = vy * d.isy;
    }
    const unsigned short* wg = (const unsigned short*)a.w;
    const int kgroups = a.ci_pad >> 3;
    const long plane_elems = (long)d.w_taps * kgroups * a.co_pad * 8;
    int wn[NJ];
    bool okn[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int n = n0 + j * 32 + r;
        okn[j] = n < a.co_pad;
        wn[j] = (okn[j] ? n : 0) * 8;
    }
    const int aff = b * d.in_affine_batch_stride;
    __syncthreads();
    struct Frag {
        uint4 x[MI][X3 ? 2 : 1];
        float4 sc[2], sh[2];
        uint4 bh[NJ], bl[X3 ? NJ : 1];
        unsigned ok;  // bit i: the pixel of tile row i exists; bits 8 / 9: the lane's first / second 4 channels exist
    };
    // the ISSUE cursor: (tap, first channel of the step); beyond the last step it stays (the last step again, never multiplied)
    const int K = a.ci_pad;  // multiple of 16
    const int n_steps = (tap1 - tap0) * (K >> 4);
    int cur_tap = tap0, cur_k = 0;
    auto load = [&](Frag& F) {
        const int c = cur_k + 8 * h;
        const bool v0 = c < d.ci, v1 = c + 4 < d.ci;  // (fp32: ci is a multiple of 4, bf16: of 8 -- a 16-byte chunk is whole or absent)
        const int c_lo = v0 ? c : 0, c_hi = v1 ? c + 4 : 0;
        const int dy = s_dy[cur_tap], dx = s_dx[cur_tap];
        const int ix = bx + dx;
        unsigned ok = (v0 ? 256u : 0u) | (v1 ? 512u : 0u);
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const int iy = by[i] + dy;
            const bool in = okv[i] && (unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi;
            ok |= in ? (1u << i) : 0u;
            const unsigned char* px = xb + (in ? ((long)iy * d.wi + ix) * d.x_pix_stride : 0) * ES;
            F.x[i][0] = *reinterpret_cast<const uint4*>(px + c_lo * ES);
            if constexpr (X3) F.x[i][1] = *reinterpret_cast<const uint4*>(px + c_hi * ES);
        }
        F.ok = ok;
        if constexpr (PRO) {
            F.sc[0] = *reinterpret_cast<const float4*>(a.in_scale + aff + c_lo);
            F.sc[1] = *reinterpret_cast<const float4*>(a.in_scale + aff + c_hi);
            F.sh[0] = *reinterpret_cast<const float4*>(a.in_shift + aff + c_lo);
            F.sh[1] = *reinterpret_cast<const float4*>(a.in_shift + aff + c_hi);
        }
        const long ko = ((long)s_w[cur_tap] * kgroups + (cur_k >> 3) + h) * a.co_pad * 8;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            F.bh[j] = *reinterpret_cast<const uint4*>(wg + ko + wn[j]);
            if constexpr (X3) F.bl[j] = *reinterpret_cast<const uint4*>(wg + plane_elems + ko + wn[j]);
        }
        // advance (VALU only)
        if (cur_k + 16 < K) {
            cur_k += 16;
        } else if (cur_tap + 1 < tap1) {
            cur_tap++;
            cur_k = 0;
        }
    };
    f16v acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    auto mul = [&](const Frag& F) {
        const bool v0 = (F.ok >> 8) & 1u, v1 = (F.ok >> 9) & 1u;
        uint4 ah[MI], al[X3 ? MI : 1];
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const bool in = (F.ok >> i) & 1u;
            float f[8];
            if constexpr (X3) {
                const uint4 q0 = F.x[i][0], q1 = F.x[i][1];
                const unsigned w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int e = 0; e < 8; e++) f[e] = __uint_as_float(w[e]);
            } else {
                const uint4 q0 = F.x[i][0];
                const unsigned w[4] = {q0.x, q0.y, q0.z, q0.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    f[2 * e] = bf16_lo(w[e]);
                    f[2 * e + 1] = bf16_hi(w[e]);
                }
            }
            if constexpr (PRO) {
                const float sc[8] = {F.sc[0].x, F.sc[0].y, F.sc[0].z, F.sc[0].w, F.sc[1].x, F.sc[1].y, F.sc[1].z, F.sc[1].w};
                const float sh[8] = {F.sh[0].x, F.sh[0].y, F.sh[0].z, F.sh[0].w, F.sh[1].x, F.sh[1].y, F.sh[1].z, F.sh[1].w};
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    f[e] = fmaf(f[e], sc[e], sh[e]);
                    if (d.in_relu) f[e] = fmaxf(f[e], 0.0f);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (!(in && (e < 4 ? v0 : v1))) f[e] = 0.0f;
            unsigned hi[4];
#pragma unroll
            for (int e = 0; e < 4; e++) hi[e] = pack_bf16(f[2 * e], f[2 * e + 1]);
            ah[i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            if constexpr (X3) {
                unsigned lo[4];
#pragma unroll
                for (int e = 0; e < 4; e++) lo[e] = pack_bf16(f[2 * e] - bf16_lo(hi[e]), f[2 * e + 1] - bf16_hi(hi[e]));
                al[i] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint4 bh = okn[j] ? F.bh[j] : make_uint4(0u, 0u, 0u, 0u);
            uint4 bl = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (X3) bl = okn[j] ? F.bl[j] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int i = 0; i < MI; i++) {
                if constexpr (X3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(al[i]), as_bf8(bh), acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah[i]), as_bf8(bl), acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah[i]), as_bf8(bh), acc[i][j], 0, 0, 0);
            }
        }
    };
    Frag F0, F1;
    load(F0);
    for (int s_ = 0; s_ < n_steps; s_ += 2) {
        load(F1);
        __builtin_amdgcn_sched_barrier(0);
        mul(F0);
        __builtin_amdgcn_sched_barrier(0);
        if (s_ + 1 < n_steps) {
            load(F0);
            __builtin_amdgcn_sched_barrier(0);
            mul(F1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    conv_epilogue<MI, NJ, OUT_F32>(d, a, acc, cls, b, tx, ty, wave, r, h, true, n0, stats_row, tid_all, smem);
}

// ---- windows on 2-8 input channels, F32X3: two taps per MFMA step, fragments straight from global memory ---------------------------
// (plan_taps, conv_plan.h.)  Lane (r, h) of step s reads the <= 8 channels of pixel r SHIFTED by tap 2 s + h (zero outside the map and
// for the odd tap count's last half step) and the weight fragment of that tap (packed [tap][k8 = 0][n][8]: its channels 8-15 are the
// padding that conv_igemm_kernel multiplies).  Tile geometry, epilogue and statistics rows as conv_igemm_kernel with 4-row tiles.
template <int NJ, bool PRO>
__global__ __launch_bounds__(kThreads, 2) void conv_taps_kernel(const liso_conv_desc d, const FwdArgs a) {
    constexpr int BNT = 32 * NJ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 96 * 2 * 4];
    __shared__ int s_dy[LISO_CONV_MAX_TAPS + 1], s_dx[LISO_CONV_MAX_TAPS + 1], s_w[LISO_CONV_MAX_TAPS + 1];
    const int tid_all = threadIdx.x, wave = tid_all >> 6, lane = tid_all & 63, r = lane & 31, h = lane >> 5;
    if (tid_all <= d.n_taps && tid_all <= LISO_CONV_MAX_TAPS) {  // (entry n_taps: the odd count's missing tap -- far outside every map)
        const bool real = tid_all < d.n_taps;
        s_dy[tid_all] = real ? d.tap_dy[tid_all] : (1 << 28);
        s_dx[tid_all] = real ? d.tap_dx[tid_all] : (1 << 28);
        s_w[tid_all] = real ? d.tap_w[tid_all] : 0;
    }
    int t = xcd_remap(blockIdx.x, a.total);
    const int nt = t % a.n_nt;
    t /= a.n_nt;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    const int stats_row = (b * a.tiles_y + ty) * a.tiles_x + tx;
    const int n0 = nt * BNT;
    const int vy = ty * 4 + wave, vx = tx * 32 + r;
    const bool okv = vy < d.hv && vx < d.wv;
    const int by = vy * d.isy, bx = vx * d.isx;
    const float* xb = (const float*)a.x + (long)b * d.hi * d.wi * d.x_pix_stride;
    const unsigned short* wg = (const unsigned short*)a.w;
    const int kgroups = a.ci_pad >> 3;
    const long plane_elems = (long)d.w_taps * kgroups * a.co_pad * 8;
    int wn[NJ];
    bool okn[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        const int n = n0 + j * 32 + r;
        okn[j] = n < a.co_pad;
        wn[j] = (okn[j] ? n : 0) * 8;
    }
    const bool c1 = d.ci > 4;  // (ci is a multiple of 4: the second 16-byte chunk of the lane's 8 channels exists or not)
    float sc[8], sh[8];
    if constexpr (PRO) {
        const int aff = b * d.in_affine_batch_stride;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const bool ok = e < d.ci;
            sc[e] = ok ? a.in_scale[aff + e] : 0.0f;
            sh[e] = ok ? a.in_shift[aff + e] : 0.0f;
        }
    }
    __syncthreads();
    struct Frag {
        float4 x[2];
        uint4 bh[NJ], bl[NJ];
        bool ok;
    };
    const int NS = (d.n_taps + 1) >> 1;
    auto load = [&](int s_, Frag& F) {
        const int tap = 2 * s_ + h;
        const int iy = by + s_dy[tap], ix = bx + s_dx[tap];
        F.ok = okv && (unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi;
        const float* px = xb + (F.ok ? ((long)iy * d.wi + ix) * d.x_pix_stride : 0);
        F.x[0] = *reinterpret_cast<const float4*>(px);
        F.x[1] = *reinterpret_cast<const float4*>(px + (c1 ? 4 : 0));
        const long ko = (long)s_w[tap] * kgroups * a.co_pad * 8;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            F.bh[j] = *reinterpret_cast<const uint4*>(wg + ko + wn[j]);
            F.bl[j] = *reinterpret_cast<const uint4*>(wg + plane_elems + ko + wn[j]);
        }
    };
    f16v acc[1][NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[0][j][e] = 0.0f;
    auto mul = [&](int s_, const Frag& F) {
        const bool real = 2 * s_ + h < d.n_taps;
        float f[8] = {F.x[0].x, F.x[0].y, F.x[0].z, F.x[0].w, F.x[1].x, F.x[1].y, F.x[1].z, F.x[1].w};
        if constexpr (PRO) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                f[e] = fmaf(f[e], sc[e], sh[e]);
                if (d.in_relu) f[e] = fmaxf(f[e], 0.0f);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++)
            if (!(F.ok && real && (e < 4 || c1))) f[e] = 0.0f;
        unsigned hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            hi[e] = pack_bf16(f[2 * e], f[2 * e + 1]);
            lo[e] = pack_bf16(f[2 * e] - bf16_lo(hi[e]), f[2 * e + 1] - bf16_hi(hi[e]));
        }
        const uint4 ah = make_uint4(hi[0], hi[1], hi[2], hi[3]), al = make_uint4(lo[0], lo[1], lo[2], lo[3]);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const bool okb = okn[j] && real;
            const uint4 bh = okb ? F.bh[j] : make_uint4(0u, 0u, 0u, 0u), bl = okb ? F.bl[j] : make_uint4(0u, 0u, 0u, 0u);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(al), as_bf8(bh), acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah), as_bf8(bl), acc[0][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf8(ah), as_bf8(bh), acc[0][j], 0, 0, 0);
        }
    };
    // three steps in the air: the loads of steps s + 1 and s + 2 are in flight under the MFMAs of step s (beyond the end: the last
    // step again, never multiplied -- no branch around a load)
    Frag F0, F1, F2;
    auto at = [&](int s_) { return s_ < NS ? s_ : NS - 1; };
    load(0, F0);
    load(at(1), F1);
    for (int s_ = 0; s_ < NS; s_ += 3) {
        load(at(s_ + 2), F2);
        __builtin_amdgcn_sched_barrier(0);
        mul(s_, F0);
        __builtin_amdgcn_sched_barrier(0);
        if (s_ + 1 < NS) {
            load(at(s_ + 3), F0);
            __builtin_amdgcn_sched_barrier(0);
            mul(s_ + 1, F1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s_ + 2 < NS) {
            load(at(s_ + 4), F1);
            __builtin_amdgcn_sched_barrier(0);
            mul(s_ + 2, F2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    conv_epilogue<1, NJ, true>(d, a, acc, 0, b, tx, ty, wave, r, h, true, n0, stats_row, tid_all, smem);
}

// ---- weight packing ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_chunk(const float* __restrict__ src, int d1, int taps, int swap_ab, int K, int N, int Kp, int Np,
                                           int f32, unsigned short* __restrict__ dst, long q);

// 16-B chunks of the packed weights: bf16 planes x taps x Kp/8 x Np; exact fp32 (one plane of 4-float groups): taps x Kp/4 x Np
__host__ __device__ inline long pack_chunks(int planes, int taps, int Kp, int Np, int f32) {
    return f32 ? (long)taps * (Kp / 4) * Np : (long)planes * taps * (Kp / 8) * Np;
}

__global__ void pack_weights_kernel(const float* __restrict__ src, int d0, int d1, int taps, int swap_ab, int K, int N, int Kp,
                                    int Np, int planes, int f32, unsigned short* __restrict__ dst) {
    const long total = pack_chunks(planes, taps, Kp, Np, f32);  // one thread per 16-B chunk
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    pack_chunk(src, d1, taps, swap_ab, K, N, Kp, Np, f32, dst, q);
}

// several weight tensors in one launch (a training step packs every layer twice: forward and data-gradient panels)
constexpr int kPackJobs = 48;
struct PackJob {
    const float* src;
    unsigned short* dst;
    int d1, taps, swap_ab, K, N, Kp, Np, f32;
};
struct PackTable {
    PackJob job[kPackJobs];
    long end[kPackJobs];  // exclusive prefix of chunk counts
    int n;
};

__global__ void pack_weights_batched_kernel(const PackTable t) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= t.end[t.n - 1]) return;
    int j = 0;
    while (q >= t.end[j]) j++;
    const PackJob& b = t.job[j];
    pack_chunk(b.src, b.d1, b.taps, b.swap_ab, b.K, b.N, b.Kp, b.Np, b.f32, b.dst, q - (j ? t.end[j - 1] : 0));
}

__device__ __forceinline__ void pack_chunk(const float* __restrict__ src, int d1, int taps, int swap_ab, int K, int N, int Kp, int Np,
                                           int f32, unsigned short* __restrict__ dst, long q) {
    const int n = (int)(q % Np);
    long t = q / Np;
    if (f32) {  // [tap][Kp / 4][Np][4] fp32, unrounded
        const int k4 = (int)(t % (Kp / 4));
        const int tap = (int)(t / (Kp / 4));
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int k = k4 * 4 + e;
            f[e] = 0.0f;
            if (k < K && n < N) {
                const int ia = swap_ab ? k : n, ib = swap_ab ? n : k;
                f[e] = src[((long)ia * d1 + ib) * taps + tap];
            }
        }
        *reinterpret_cast<float4*>(dst + q * 8) = make_float4(f[0], f[1], f[2], f[3]);
        return;
    }
    const int k8 = (int)(t % (Kp / 8));
    t /= (Kp / 8);
    const int tap = (int)(t % taps);
    const int plane = (int)(t / taps);
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        float f[2];
#pragma unroll
        for (int z = 0; z < 2; z++) {
            const int k = k8 * 8 + 2 * e + z;
            float v = 0.0f;
            if (k < K && n < N) {
                const int ia = swap_ab ? k : n, ib = swap_ab ? n : k;  // src[ia][ib][tap]
                v = src[((long)ia * d1 + ib) * taps + tap];
            }
            const float hi = round_bf16(v);
            f[z] = plane == 0 ? hi : (v - hi);
        }
        w[e] = pack_bf16(f[0], f[1]);
    }
    *reinterpret_cast<uint4*>(dst + q * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- BatchNorm statistics from the per-block partial sums ---------------------------------------------------------------------
constexpr int kFinGroups = 64;  // row groups of the finalize kernel (1024 threads = 64 groups x 8 channels x 2 sums)
__global__ __launch_bounds__(1024) void conv_bn_finalize_kernel(const float* __restrict__ partial, int rows, int co, int co_pad,
                                                                long n, const float* __restrict__ stats_shift,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                float momentum, float eps, float* __restrict__ stats) {
    // 8 channels per block; 64 row groups walk the per-block partial sums (1024-2048 rows of 8 B at a 512-B stride: latency-bound --
    // with 16 groups the kernel took 7-11 us, a fifth of the convolution in front of it), fp64 merge in a fixed order
    __shared__ double red[kFinGroups][16];
    // blockIdx.y = sample (InstanceNorm: one statistics row set per sample; BatchNorm launches a single y)
    partial += (size_t)blockIdx.y * rows * 2 * co_pad;
    stats += (size_t)blockIdx.y * 4 * co;
    const int col = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 8 + (col & 7), s = col >> 3;
    double acc = 0.0;
    if (c < co) {
        float v[16];
        int row = rg;
        for (; row + 15 * kFinGroups < rows; row += 16 * kFinGroups) {  // 16 independent loads in flight per thread, summed in row order
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = partial[((long)(row + u * kFinGroups) * 2 + s) * co_pad + c];
#pragma unroll
            for (int u = 0; u < 16; u++) acc += (double)v[u];
        }
        for (; row + 3 * kFinGroups < rows; row += 4 * kFinGroups) {
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = partial[((long)(row + u * kFinGroups) * 2 + s) * co_pad + c];
#pragma unroll
            for (int u = 0; u < 4; u++) acc += (double)v[u];
        }
        for (; row < rows; row += kFinGroups) acc += (double)partial[((long)row * 2 + s) * co_pad + c];
    }
    red[rg][col] = acc;
    __syncthreads();
    // the 64 group sums per (channel, sum) as a fixed two-level tree: 8 runs of 8 consecutive groups, then the 8 run sums in order (64
    // dependent LDS reads + fp64 adds in ONE thread were ~3 us of this 5.6-us launch)
    double q = 0.0;
    if (threadIdx.x < 128) {
        const int run = threadIdx.x >> 4;
#pragma unroll
        for (int g = 0; g < 8; g++) q += red[run * 8 + g][col];
    }
    __syncthreads();
    if (threadIdx.x < 128) red[threadIdx.x >> 4][col] = q;
    __syncthreads();
    if (threadIdx.x < 16) {
        q = 0.0;
#pragma unroll
        for (int g = 0; g < 8; g++) q += red[g][threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x < 16) red[0][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < 8) {
        const int cc = blockIdx.x * 8 + threadIdx.x;
        if (cc < co) {
            const double q1 = red[0][threadIdx.x], q2 = red[0][8 + threadIdx.x];
            const double k = stats_shift ? (double)stats_shift[cc] : 0.0;
            const double m1 = q1 / (double)n;
            double var = q2 / (double)n - m1 * m1;
            if (var < 0.0) var = 0.0;
            const double mean = k + m1;
            const double invstd = 1.0 / sqrt(var + (double)eps);
            const double sc = (gamma ? (double)gamma[cc] : 1.0) * invstd;
            stats[cc] = (float)sc;
            stats[co + cc] = (float)((beta ? (double)beta[cc] : 0.0) - mean * sc);
            stats[2 * co + cc] = (float)mean;
            stats[3 * co + cc] = (float)invstd;
            if (running_mean) {
                const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
                running_mean[cc] = (float)((1.0 - momentum) * running_mean[cc] + momentum * mean);
                running_var[cc] = (float)((1.0 - momentum) * running_var[cc] + momentum * unb);
            }
        }
    }
}

// ---- tail of a residual block: out = relu(fa(a) + fb(b)), pending per-sample affine (+ ReLU) of either branch applied on the fly ----
__global__ __launch_bounds__(256) void residual_affine_relu_kernel(const float4* __restrict__ a, const float* __restrict__ a_scale,
                                                                   const float* __restrict__ a_shift, int a_relu,
                                                                   const float4* __restrict__ b, const float* __restrict__ b_scale,
                                                                   const float* __restrict__ b_shift, int b_relu,
                                                                   float4* __restrict__ out, long per_sample4, int c4, long total4, int a_stride4,
                                                                   int b_stride4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const long n = i / per_sample4;
        const int cq = (int)(i % c4);  // (per_sample4 is a multiple of c4)
        float4 va = a[i], vb = b[i];
        if (a_scale) {
            const float4 s = reinterpret_cast<const float4*>(a_scale)[n * a_stride4 + cq], t = reinterpret_cast<const float4*>(a_shift)[n * a_stride4 + cq];
            va = make_float4(fmaf(va.x, s.x, t.x), fmaf(va.y, s.y, t.y), fmaf(va.z, s.z, t.z), fmaf(va.w, s.w, t.w));
            if (a_relu) va = make_float4(fmaxf(va.x, 0.f), fmaxf(va.y, 0.f), fmaxf(va.z, 0.f), fmaxf(va.w, 0.f));
        }
        if (b_scale) {
            const float4 s = reinterpret_cast<const float4*>(b_scale)[n * b_stride4 + cq], t = reinterpret_cast<const float4*>(b_shift)[n * b_stride4 + cq];
            vb = make_float4(fmaf(vb.x, s.x, t.x), fmaf(vb.y, s.y, t.y), fmaf(vb.z, s.z, t.z), fmaf(vb.w, s.w, t.w));
            if (b_relu) vb = make_float4(fmaxf(vb.x, 0.f), fmaxf(vb.y, 0.f), fmaxf(vb.z, 0.f), fmaxf(vb.w, 0.f));
        }
        out[i] = make_float4(fmaxf(va.x + vb.x, 0.f), fmaxf(va.y + vb.y, 0.f), fmaxf(va.z + vb.z, 0.f), fmaxf(va.w + vb.w, 0.f));
    }
}

int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

template <int MODE, int MI, int NJ, bool OUT_F32, int NTAPS, bool PRO>
int launch_roles_pro(const liso_conv_desc& d, const Plan& p, hipStream_t st) {
    static liso_dev::PerDeviceFlag attr_set;  // (per device: per_device.h)
    if (!liso_dev::lds_opt_in(attr_set, (const void*)conv_roles_kernel<MODE, MI, NJ, OUT_F32, NTAPS, PRO>, 160 * 1024)) return LISO_ELAUNCH;
    const int n_cu = liso_dev::cu_count();
    if (n_cu <= 0) return LISO_ELAUNCH;
    // LISO_CONV_OPT_ROLES_CUS: leave compute units to the other streams' kernels (include/liso_conv.h)
    const int cus = (g_roles_cus >= 8 && g_roles_cus < n_cu) ? g_roles_cus : n_cu;
    const int grid = p.a.total < cus ? p.a.total : cus;
    conv_roles_kernel<MODE, MI, NJ, OUT_F32, NTAPS, PRO><<<grid, 512, p.lds, st>>>(d, p.a);
    return check_launch();
}
template <int MODE, int MI, int NJ, bool OUT_F32, int NTAPS>
int launch_roles(const liso_conv_desc& d, const Plan& p, hipStream_t st) {
    return p.a.in_scale ? launch_roles_pro<MODE, MI, NJ, OUT_F32, NTAPS, true>(d, p, st)
                        : launch_roles_pro<MODE, MI, NJ, OUT_F32, NTAPS, false>(d, p, st);
}

template <int MODE, int MI, int NJ, bool OUT_F32>
int launch_1x1(const liso_conv_desc& d, const Plan& p, hipStream_t st) {
    if (p.a.in_scale)
        conv_1x1_kernel<MODE, MI, NJ, OUT_F32, true><<<p.a.total, kThreads, 0, st>>>(d, p.a);
    else
        conv_1x1_kernel<MODE, MI, NJ, OUT_F32, false><<<p.a.total, kThreads, 0, st>>>(d, p.a);
    return check_launch();
}

template <int NJ>
int launch_taps(const liso_conv_desc& d, const Plan& p, hipStream_t st) {
    if (p.a.in_scale)
        conv_taps_kernel<NJ, true><<<p.a.total, kThreads, 0, st>>>(d, p.a);
    else
        conv_taps_kernel<NJ, false><<<p.a.total, kThreads, 0, st>>>(d, p.a);
    return check_launch();
}

template <int MODE, int MI, int NJ, bool OUT_F32, int CS, int SK = 1>
int launch(const liso_conv_desc& d, const Plan& p, hipStream_t st) {
    static liso_dev::PerDeviceFlag attr_set;
    if (!liso_dev::lds_opt_in(attr_set, (const void*)conv_igemm_kernel<MODE, MI, NJ, OUT_F32, CS, SK>, 160 * 1024)) return LISO_ELAUNCH;
    conv_igemm_kernel<MODE, MI, NJ, OUT_F32, CS, SK><<<p.a.total, kThreads * SK, p.lds, st>>>(d, p.a);
    return check_launch();
}

}  // namespace

extern "C" int liso_conv_set_option(int option, int value) {
    if (option == LISO_CONV_OPT_SHARED_GPU) {
        g_shared_gpu = value != 0;
        return LISO_OK;
    }
    if (option == LISO_CONV_OPT_ROLES_CUS) {
        if (value != 0 && value < 8) return LISO_EINVAL;
        g_roles_cus = value;
        return LISO_OK;
    }
    return LISO_EINVAL;
}

extern "C" {

size_t liso_conv_packed_bytes(int k_channels, int n_channels, int taps, int mode) {
    if (k_channels <= 0 || n_channels <= 0 || taps <= 0) return 0;
    const size_t planes = mode == LISO_CONV_BF16 ? 1 : 2;  // (exact fp32: 4 B per element = the bytes of two bf16 planes)
    return planes * (size_t)taps * round_up(k_channels, 16) * round_up(n_channels, 64) * 2;
}

int liso_conv_pack_weights(const float* src, int d0, int d1, int kh, int kw, int transposed, int for_dgrad, int mode, void* dst,
                           void* stream) {
    if (!src || !dst || d0 <= 0 || d1 <= 0 || kh <= 0 || kw <= 0) return LISO_EINVAL;
    const bool same = (transposed != 0) == (for_dgrad != 0);
    const int K = same ? d1 : d0, N = same ? d0 : d1;
    const int Kp = round_up(K, 16), Np = round_up(N, 64), taps = kh * kw, planes = mode == LISO_CONV_F32X3 ? 2 : 1;
    const int f32 = mode == LISO_CONV_F32;
    const long total = pack_chunks(planes, taps, Kp, Np, f32);
    pack_weights_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(src, d0, d1, taps, same ? 0 : 1, K, N, Kp, Np,
                                                                                      planes, f32, (unsigned short*)dst);
    return check_launch();
}

int liso_conv_pack_weights_batched(const liso_conv_pack_job* jobs, int n_jobs, void* stream) {
    if (n_jobs == 0) return LISO_OK;
    if (!jobs || n_jobs < 0) return LISO_EINVAL;
    for (int base = 0; base < n_jobs; base += kPackJobs) {
        PackTable t;
        t.n = n_jobs - base < kPackJobs ? n_jobs - base : kPackJobs;
        long run = 0;
        for (int i = 0; i < t.n; i++) {
            const liso_conv_pack_job& j = jobs[base + i];
            if (!j.src || !j.dst || j.d0 <= 0 || j.d1 <= 0 || j.kh <= 0 || j.kw <= 0) return LISO_EINVAL;
            const bool same = (j.transposed != 0) == (j.for_dgrad != 0);
            const int K = same ? j.d1 : j.d0, N = same ? j.d0 : j.d1;
            const int Kp = round_up(K, 16), Np = round_up(N, 64), taps = j.kh * j.kw, planes = j.mode == LISO_CONV_F32X3 ? 2 : 1;
            const int f32 = j.mode == LISO_CONV_F32;
            t.job[i] = PackJob{j.src, (unsigned short*)j.dst, j.d1, taps, same ? 0 : 1, K, N, Kp, Np, f32};
            run += pack_chunks(planes, taps, Kp, Np, f32);
            t.end[i] = run;
        }
        pack_weights_batched_kernel<<<(unsigned)((run + 255) / 256), 256, 0, (hipStream_t)stream>>>(t);
    }
    return check_launch();
}

int liso_conv_kernel_kind(const liso_conv_desc* d) {
    Plan p;
    if (!d || !make_plan(*d, &p)) return -1;
    return p.a.roles ? 1 : p.a.direct1x1 ? 2 : p.a.direct_taps ? 3 : 0;
}

int liso_conv_plan_info(const liso_conv_desc* d, int info[8]) {
    Plan p;
    if (!d || !info || !make_plan(*d, &p)) return LISO_EINVAL;
    info[0] = p.a.roles ? 1 : p.a.direct1x1 ? 2 : p.a.direct_taps ? 3 : 0;
    info[1] = p.mi;
    info[2] = p.nj;
    info[3] = p.sk;
    info[4] = p.a.total;
    info[5] = p.lds;
    info[6] = p.cs;
    info[7] = p.a.roles ? 9 : p.g;
    return LISO_OK;
}

int liso_conv_stats_rows(const liso_conv_desc* d) {
    Plan p;
    if (!d || !make_plan(*d, &p)) return -1;
    return d->n_classes * d->batch * p.a.tiles_y * p.a.tiles_x * (p.a.roles ? p.mi : 1);  // (conv_roles_kernel: one row per 4 x 32 pixels)
}

int liso_conv_forward(const liso_conv_desc* d, const void* x, const void* w_packed, const float* bias, const float* in_scale,
                      const float* in_shift, void* y, float* stats_partial, const float* stats_shift, void* stream) {
    return liso_conv_forward_sparse(d, x, w_packed, bias, in_scale, in_shift, y, stats_partial, stats_shift, nullptr, stream);
}

int liso_conv_forward_sparse(const liso_conv_desc* d, const void* x, const void* w_packed, const float* bias, const float* in_scale,
                             const float* in_shift, void* y, float* stats_partial, const float* stats_shift,
                             const float* occupancy, void* stream) {
    if (!d || !x || !w_packed || !y) return LISO_EINVAL;
    if (occupancy && in_scale) return LISO_EINVAL;  // (a prologue maps zeros to relu(shift): the tile is no longer zero)
    if ((in_scale == nullptr) != (in_shift == nullptr)) return LISO_EINVAL;
    if (((uintptr_t)x | (uintptr_t)w_packed) & 15) return LISO_EINVAL;
    Plan p;
    if (!make_plan(*d, &p)) return LISO_EINVAL;
    p.a.x = x;
    p.a.w = w_packed;
    p.a.bias = bias;
    p.a.in_scale = in_scale;
    p.a.in_shift = in_shift;
    p.a.y = y;
    p.a.stats = stats_partial;
    p.a.stats_shift = stats_shift;
    p.a.occ = occupancy;
    hipStream_t st = (hipStream_t)stream;
    const bool x3 = d->mode == LISO_CONV_F32X3;
    const bool of32 = x3 || d->out_f32;
    const bool f32 = d->mode == LISO_CONV_F32;
#define LISO_GO(MODE, MI, NJ, OF, CSA, CSB) return p.cs == CSA ? launch<MODE, MI, NJ, OF, CSA>(*d, p, st) : launch<MODE, MI, NJ, OF, CSB>(*d, p, st)
#define LISO_SEL(MODE, OF, CSA, CSB)                  \
    do {                                    \
        if (p.mi == 2 && p.nj == 2) LISO_GO(MODE, 2, 2, OF, CSA, CSB); \
        if (p.mi == 2 && p.nj == 1) LISO_GO(MODE, 2, 1, OF, CSA, CSB); \
        if (p.mi == 1 && p.nj == 2) LISO_GO(MODE, 1, 2, OF, CSA, CSB); \
        LISO_GO(MODE, 1, 1, OF, CSA, CSB);            \
    } while (0)
    if (p.a.roles) {
        // (an occupancy map is ignored here: the dense result is bit-identical by contract)
#define LISO_ROLES(MODE, OF)                                                                   \
    do {                                                                                       \
        if (p.mi == 1 && p.nj == 1) return launch_roles<MODE, 1, 1, OF, 9>(*d, p, st);         \
        if (p.mi == 1 && p.nj == 2) return launch_roles<MODE, 1, 2, OF, 9>(*d, p, st);         \
        if (p.mi == 2 && p.nj == 1) return launch_roles<MODE, 2, 1, OF, 9>(*d, p, st);         \
        if (p.mi == 2 && p.nj == 2) return launch_roles<MODE, 2, 2, OF, 9>(*d, p, st);         \
    } while (0)
        if (x3) {
            if (p.mi == 1 && p.nj == 3) return launch_roles<LISO_CONV_F32X3, 1, 3, true, 9>(*d, p, st);
            LISO_ROLES(LISO_CONV_F32X3, true);
        } else if (of32) {
            LISO_ROLES(LISO_CONV_BF16, true);
        } else {
            LISO_ROLES(LISO_CONV_BF16, false);
        }
#undef LISO_ROLES
        return LISO_EINVAL;
    }
    if (p.a.direct1x1) {
        if (((uintptr_t)in_scale | (uintptr_t)in_shift) & 15) return LISO_EINVAL;
#define LISO_D1(MODE, OF)                                                                  \
    do {                                                                                   \
        if (p.mi == 1 && p.nj == 1) return launch_1x1<MODE, 1, 1, OF>(*d, p, st);          \
        if (p.mi == 1 && p.nj == 2) return launch_1x1<MODE, 1, 2, OF>(*d, p, st);          \
        if (p.mi == 1 && p.nj == 3) return launch_1x1<MODE, 1, 3, OF>(*d, p, st);          \
        if (p.mi == 2 && p.nj == 1) return launch_1x1<MODE, 2, 1, OF>(*d, p, st);          \
        if (p.mi == 2 && p.nj == 2) return launch_1x1<MODE, 2, 2, OF>(*d, p, st);          \
        if (p.mi == 2 && p.nj == 3) return launch_1x1<MODE, 2, 3, OF>(*d, p, st);          \
    } while (0)
        if (x3) LISO_D1(LISO_CONV_F32X3, true);
        else if (of32) LISO_D1(LISO_CONV_BF16, true);
        else LISO_D1(LISO_CONV_BF16, false);
#undef LISO_D1
        return LISO_EINVAL;
    }
    if (p.a.direct_taps) {
        if (p.nj == 1) return launch_taps<1>(*d, p, st);
        if (p.nj == 2) return launch_taps<2>(*d, p, st);
        if (p.nj == 3) return launch_taps<3>(*d, p, st);
        return LISO_EINVAL;
    }
    if (x3 && p.sk == 2)
        return p.cs == 32 ? launch<LISO_CONV_F32X3, 1, 1, true, 32, 2>(*d, p, st) : launch<LISO_CONV_F32X3, 1, 1, true, 16, 2>(*d, p, st);
    if (f32) LISO_SEL(LISO_CONV_F32, true, 32, 16);
    if (x3 && p.nj == 3) {
        if (p.mi == 2) LISO_GO(LISO_CONV_F32X3, 2, 3, true, 32, 16);
        LISO_GO(LISO_CONV_F32X3, 1, 3, true, 32, 16);
    }
    if (x3) LISO_SEL(LISO_CONV_F32X3, true, 32, 16);
    if (of32) LISO_SEL(LISO_CONV_BF16, true, 64, 32);
    LISO_SEL(LISO_CONV_BF16, false, 64, 32);
#undef LISO_SEL
#undef LISO_GO
}

int liso_conv_bn_finalize(const float* stats_partial, int rows, int co, int co_pad, long n, const float* stats_shift,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, float* stats, void* stream) {
    if (!stats_partial || !gamma || !beta || !stats || rows <= 0 || co <= 0 || co_pad < co || n <= 0) return LISO_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return LISO_EINVAL;
    conv_bn_finalize_kernel<<<(co + 7) / 8, 1024, 0, (hipStream_t)stream>>>(stats_partial, rows, co, co_pad, n, stats_shift, gamma,
                                                                          beta, running_mean, running_var, momentum, eps, stats);
    return check_launch();
}

int liso_conv_in_finalize(const float* stats_partial, int rows_per_sample, int batch, int co, int co_pad, long n_per_sample,
                          const float* gamma, const float* beta, float eps, float* stats, void* stream) {
    if (!stats_partial || !stats || rows_per_sample <= 0 || batch <= 0 || co <= 0 || co_pad < co || n_per_sample <= 0) return LISO_EINVAL;
    if ((gamma == nullptr) != (beta == nullptr)) return LISO_EINVAL;
    conv_bn_finalize_kernel<<<dim3((co + 7) / 8, batch), 1024, 0, (hipStream_t)stream>>>(stats_partial, rows_per_sample, co, co_pad,
                                                                                     n_per_sample, nullptr, gamma, beta, nullptr,
                                                                                     nullptr, 0.0f, eps, stats);
    return check_launch();
}

int liso_residual_affine_relu_f32(const float* a, const float* a_scale, const float* a_shift, int a_stride, int a_relu, const float* b,
                                  const float* b_scale, const float* b_shift, int b_stride, int b_relu, float* out, int batch,
                                  long pixels, int c, void* stream) {
    if (!a || !b || !out || batch <= 0 || pixels <= 0 || c <= 0 || (c & 3) || (a_stride & 3) || (b_stride & 3)) return LISO_EINVAL;
    if ((a_scale == nullptr) != (a_shift == nullptr) || (b_scale == nullptr) != (b_shift == nullptr)) return LISO_EINVAL;
    if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) != 0) return LISO_EINVAL;
    const long per4 = pixels * (c / 4), total4 = per4 * batch;
    long blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    residual_affine_relu_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float4*>(a), a_scale, a_shift, a_relu, reinterpret_cast<const float4*>(b), b_scale, b_shift, b_relu,
        reinterpret_cast<float4*>(out), per4, c / 4, total4, a_stride / 4, b_stride / 4);
    return check_launch();
}

}  // extern "C"
