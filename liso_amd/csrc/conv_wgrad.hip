// Weight gradient of the NHWC convolutions on the gfx950 matrix cores (C ABI: include/liso_conv.h, liso_conv_wgrad).
//
//   dW[tap][ci][co] = sum over (sample, virtual pixel v) of  x'[v*is + tap][ci] * dy[v*os + oo][co]
// is a GEMM whose reduction axis is the PIXEL axis, while both operands are stored channel-contiguous (NHWC).  The MFMA
// wants 8 consecutive k per lane, i.e. 8 consecutive pixels of one channel: the tiles are staged into LDS exactly as they
// lie in memory ([pixel][channel], the same halo tile + BatchNorm/ReLU prologue as the forward kernel) and read back with
// ds_read_b64_tr_b16, gfx950's transposing LDS read (a 16-lane group reads 4 pixels x 16 channels and receives them
// channel-major): no shuffles, no transposed copy.
//
// Block = 256 threads = 4 waves, one (64 input channels x 64 output channels) tile of up to TG taps; wave w owns the
// 32 x 32 quadrant (ci half w >> 1, co half w & 1) of every tap -> TG accumulators of 16 registers.  The block walks its
// share of the 4 x 32-pixel tiles of the batch (split-K over pixels), keeps the sums in registers and writes ONE fp32
// slab at the end; a second kernel adds the slabs in a fixed order (no float atomics: bitwise reproducible) and emits the
// gradient in torch's weight layout, and the bias gradient (column sums of dy, accumulated while dy is staged).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liso_conv.h"
#include "../../include/liso_iou3d.h"
#include "per_device.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;

constexpr int kThreads = 256;
constexpr int TW = 32;  // virtual pixels per tile row; 4, 2 or 1 rows per tile (WgArgs::th), fewer when the halo tile is large
constexpr int CT = 64;                           // channel tile (both ci and co)
// LDS bytes per pixel of a bf16 tile (x and dy): 64 channels + 64 B.  The transposing reads of one 32-lane group touch 4 consecutive
// pixels x 64 B: with this stride (48 banks) the four 16-bank ranges tile the 64 banks exactly -- the former 144-B stride (36 banks)
// made pixels 0 / 2 and 1 / 3 overlap by 8 banks: 36 % of all LDS cycles were bank conflicts (profiles/r03_detector_pmc_LDS.csv)
constexpr int PSB = CT * 2 + 64;
constexpr int PS32 = CT * 4 + 16;                // LDS bytes per pixel of an fp32 tile (exact mode: x and dy)

struct WgArgs {
    const void* x;
    const float* in_scale;
    const float* in_shift;
    const void* dy;
    float* slab;       // [splits][w_taps][ci_t * 64][co_t * 64]
    float* bias_slab;  // [classes][splits][co_t * 64] or null
    int dy_pix_stride;
    int ci_t, co_t;    // channel tiles
    int n_groups;      // tap groups over all classes
    int grp_cls[LISO_CONV_MAX_TAPS], grp_begin[LISO_CONV_MAX_TAPS], grp_cnt[LISO_CONV_MAX_TAPS];
    int cls_dy0[LISO_CONV_MAX_CLASSES], cls_dx0[LISO_CONV_MAX_CLASSES];
    int cls_inh[LISO_CONV_MAX_CLASSES], cls_inw[LISO_CONV_MAX_CLASSES];
    int tiles_x, tiles_y, n_tiles, splits;
    int x_plane_bytes;  // multiple of 16
    int psx;            // LDS bytes per x pixel
    int th;             // tile rows
    int ci_w;           // conv_wgrad_rs3_kernel: input channels per block (64, or 32: half the accumulators, half the slab per block)
    int pipelined;      // bf16: the halo tile fits one register batch -> tile k+1 is loaded during the MFMAs of tile k
    int compact;        // strided convolution whose tap groups each sit in ONE kernel row: the LDS tile holds only the input rows that
                        // group reads (row r of the tile = input row iy0 + (group's dy) + r * isy) instead of the class's dense halo
#ifdef LISO_WGRAD_STAMPS
    unsigned long long* stamps;  // diagnostic build (make STAMPS=1): 16 counters per block of conv_wgrad_rs3_kernel, scripts/wgrad_stamps.py
#endif
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float round_bf16(float v) { return (float)(__bf16)v; }

__device__ __forceinline__ bf8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
    // two transposing reads: elements 0-3 = 4 consecutive k of this lane's column, elements 4-7 the next 4
    const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)p0);
    const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)p1);
    typedef short s8 __attribute__((ext_vector_type(8)));
    const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf8, v);
}

template <int MODE, int TG>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_kernel(const liso_conv_desc d, const WgArgs a) {
    constexpr bool X3 = MODE == LISO_CONV_F32X3;
    constexpr bool F32 = MODE == LISO_CONV_F32;  // exact fp32 on v_mfma_f32_32x32x2_f32: fp32 tiles, plain 4-B LDS reads
    constexpr bool FIN = X3 || F32;
    constexpr int PLANES = X3 ? 2 : 1;
    constexpr int PSYM = F32 ? PS32 : PSB;
    constexpr int PSY = PSB;
    constexpr int XB = TG >= 9 ? 4 : 8;  // 16-B loads in flight per thread while a tile is staged (register budget)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int t = blockIdx.x;
    const int split = t % a.splits;
    t /= a.splits;
    const int cot = t % a.co_t;
    t /= a.co_t;
    const int cit = t % a.ci_t;
    const int grp = t / a.ci_t;
    const int cls = a.grp_cls[grp], tb = a.grp_begin[grp], tcnt = a.grp_cnt[grp];
    const int dy0 = a.cls_dy0[cls], dx0 = a.cls_dx0[cls], in_h = a.cls_inh[cls], in_w = a.cls_inw[cls];
    const int PSX = a.psx;
    const int npix = in_h * in_w;
    const float inv_w = 1.0f / (float)in_w;
    const int ooy = d.class_ooy[cls], oox = d.class_oox[cls];
    const int ci0 = cit * CT, co0 = cot * CT;

    unsigned char* xs = smem;
    unsigned char* ys = smem + a.x_plane_bytes * PLANES;
    const int TH = a.th, MPIX = TH * TW;
    const int y_plane = MPIX * PSYM;

    // transposing-read lane geometry (16-lane groups): group g -> k half (g >> 1), channel half-tile (g & 1)
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int ci_half = wave >> 1, co_half = wave & 1;
    const int a_lane = (8 * (g >> 1) + q) * d.isx * PSX + (ci_half * 32 + 16 * (g & 1) + 4 * p) * 2;
    const int b_lane = (8 * (g >> 1) + q) * PSY + (co_half * 32 + 16 * (g & 1) + 4 * p) * 2;

    f16v acc[TG];
#pragma unroll
    for (int i = 0; i < TG; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bsum[e] = 0.0f;
    const bool want_bias = a.bias_slab != nullptr && cit == 0 && tb == d.class_tap_begin[cls];  // once per class

    const int gdy = a.compact ? d.tap_dy[tb] - dy0 : 0;   // the group's kernel row (compact tiles start there)
    const int row_mul = a.compact ? d.isy : 1;            // input rows per tile row
    const int row_step = a.compact ? 1 : d.isy;           // tile rows per output row
    int toff[TG];
#pragma unroll
    for (int i = 0; i < TG; i++) {
        const int tp = tb + (i < tcnt ? i : 0);
        toff[i] = ((d.tap_dy[tp] - dy0 - gdy) * in_w + (d.tap_dx[tp] - dx0)) * PSX;
    }

    // ---- bf16 staging as load / store halves: the loads of tile k+1 are issued before the MFMAs of tile k (software pipeline) ----
    constexpr int YB = 4;  // dy: TH (<= 4) 16-B chunks per thread
    const int c8 = tid & 7, p0 = tid >> 3, pstep = kThreads / 8;
    uint4 xv[XB], yv[YB];
    unsigned xok = 0u, yok = 0u;
    float sc[8], sh[8];
    const bool pro = !FIN && a.in_scale != nullptr;
    if constexpr (!FIN) {
        const int ch = ci0 + c8 * 8;
        if (pro) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                sc[e] = ch < d.ci ? a.in_scale[ch + e] : 0.0f;
                sh[e] = ch < d.ci ? a.in_shift[ch + e] : 0.0f;
            }
        }
    }
    auto tile_origin = [&](int tile, int& b, int& ty, int& tx) {
        tx = tile % a.tiles_x;
        tile /= a.tiles_x;
        ty = tile % a.tiles_y;
        b = tile / a.tiles_y;
    };
    auto load_x = [&](int tile, int pix_begin) {  // XB chunks of the halo tile starting at pixel pix_begin + p0
        int b, ty, tx;
        tile_origin(tile, b, ty, tx);
        const int iy0 = ty * TH * d.isy + dy0, ix0 = tx * TW * d.isx + dx0;
        const int ch = ci0 + c8 * 8;
        const bool ch_ok = ch < d.ci;
        const unsigned short* xg = (const unsigned short*)a.x + (long)b * d.hi * d.wi * d.x_pix_stride;
        const int pfirst = pix_begin + p0;
        int ly = (int)(((float)pfirst + 0.5f) * inv_w);
        int lx = pfirst - ly * in_w;
        const int step_y = pstep / in_w, step_x = pstep - step_y * in_w;  // (uniform)
        xok = 0u;
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int pix = pfirst + u * pstep;
            const int iy = iy0 + gdy + ly * row_mul, ix = ix0 + lx;
            const bool ok = pix < npix && ch_ok && (unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi;
            xok |= ok ? (1u << u) : 0u;
            const int off = ok ? (iy * d.wi + ix) * d.x_pix_stride + ch : 0;  // (a sample has < 2^31 elements)
            lx += step_x;
            ly += step_y;
            if (lx >= in_w) {
                lx -= in_w;
                ly++;
            }
            xv[u] = *reinterpret_cast<const uint4*>(xg + off);
        }
    };
    auto store_x = [&](int pix_begin) {
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int pix = pix_begin + p0 + u * pstep;
            if (pix >= npix) continue;
            uint4 o = xv[u];
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
            if (!((xok >> u) & 1u)) o = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(xs + pix * PSX + c8 * 16) = o;
        }
    };
    auto load_y = [&](int tile) {  // TH x 32 virtual pixels x 64 channels of dy
        int b, ty, tx;
        tile_origin(tile, b, ty, tx);
        const unsigned short* yg = (const unsigned short*)a.dy + (long)b * d.ho * d.wo * a.dy_pix_stride;
        const int chy = co0 + c8 * 8;
        const bool chy_ok = chy < d.co;  // (co % 8 == 0 is required in this mode)
        yok = 0u;
#pragma unroll
        for (int u = 0; u < YB; u++) {
            if (u < TH) {
                const int m = p0 + u * pstep;
                const int vy = ty * TH + (m >> 5), vx = tx * TW + (m & 31);
                const int oy = vy * d.osy + ooy, ox = vx * d.osx + oox;
                const bool ok = chy_ok && vy < d.hv && vx < d.wv && oy < d.ho && ox < d.wo;
                yok |= ok ? (1u << u) : 0u;
                const int off = ok ? (oy * d.wo + ox) * a.dy_pix_stride + chy : 0;
                yv[u] = *reinterpret_cast<const uint4*>(yg + off);
            }
        }
    };
    auto store_y = [&]() {
#pragma unroll
        for (int u = 0; u < YB; u++) {
            if (u < TH) {
                const int m = p0 + u * pstep;
                uint4 v = yv[u];
                if (!((yok >> u) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(ys + m * PSY + c8 * 16) = v;
                if (want_bias) {
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        bsum[2 * e] += bf16_lo(w[e]);
                        bsum[2 * e + 1] += bf16_hi(w[e]);
                    }
                }
            }
        }
    };
    auto mfma_tile = [&]() {  // 8 k-steps of 16 pixels (TH = 4), all taps of the group
        if constexpr (F32) {
            // D[ci][co] += A[ci][k] B[k][co], k = 2 pixels per MFMA: lane (r = lane & 31, h = lane >> 5) holds x'[pixel 2s + h][ci r]
            // and dy[pixel 2s + h][co r] -- one 4-B LDS read each (32 consecutive channels of one pixel per lane half: conflict-free)
            const int r = lane & 31, h = lane >> 5;
            const unsigned char* bb = ys + (co_half * 32 + r) * 4;
            const unsigned char* ab = xs + (ci_half * 32 + r) * 4;
            for (int s2 = 0; s2 < MPIX / 2; s2 += 4) {
                float bv[4];
                int xo[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int m = 2 * (s2 + u) + h;
                    bv[u] = *reinterpret_cast<const float*>(bb + m * PS32);
                    xo[u] = (((m >> 5) * row_step) * in_w + (m & 31) * d.isx) * PSX;
                }
#pragma unroll
                for (int i = 0; i < TG; i++) {
                    if (i < tcnt) {
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const float av = *reinterpret_cast<const float*>(ab + xo[u] + toff[i]);
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[u], acc[i], 0, 0, 0);
                        }
                    }
                }
            }
            return;
        }
        for (int kk = 0; kk < MPIX / 16; kk++) {
            const int krow = kk >> 1, kcol = (kk & 1) * 16;
            const unsigned char* bp = ys + (krow * TW + kcol) * PSY + b_lane;
            const bf8 bh = tr_pair(bp, bp + 4 * PSY);
            bf8 bl;
            if constexpr (X3) bl = tr_pair(bp + y_plane, bp + y_plane + 4 * PSY);
            const unsigned char* ap = xs + ((krow * row_step) * in_w + kcol * d.isx) * PSX + a_lane;
#pragma unroll
            for (int i = 0; i < TG; i++) {
                if (i < tcnt) {
                    const unsigned char* api = ap + toff[i];
                    const bf8 ah = tr_pair(api, api + 4 * d.isx * PSX);
                    if constexpr (X3) {
                        const bf8 al = tr_pair(api + a.x_plane_bytes, api + a.x_plane_bytes + 4 * d.isx * PSX);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
                    }
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
                }
                // keep at most 3 taps' fragments in flight: with 9 accumulator tiles live the scheduler otherwise hoists all
                // 18 transposing reads of a k-step above the first MFMA and spills
                if (TG > 3 && (i % 3) == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (!FIN && a.pipelined) {
        // the halo tile fits one batch of XB chunks per thread: registers hold tile k+1 while the matrix cores work on tile k
        int tile = split;
        if (tile < a.n_tiles) {
            load_x(tile, 0);
            load_y(tile);
            store_x(0);
            store_y();
        }
        __syncthreads();
        for (; tile < a.n_tiles; tile += a.splits) {
            const int next = tile + a.splits;
            const bool more = next < a.n_tiles;
            if (more) {
                load_x(next, 0);
                load_y(next);
            }
            mfma_tile();
            __syncthreads();  // every wave is done with this tile's LDS image
            if (more) {
                store_x(0);
                store_y();
            }
            __syncthreads();
        }
    } else
    for (int tile = split; tile < a.n_tiles; tile += a.splits) {
        int tt = tile;
        const int tx = tt % a.tiles_x;
        tt /= a.tiles_x;
        const int ty = tt % a.tiles_y;
        const int b = tt / a.tiles_y;
        const int iy0 = ty * TH * d.isy + dy0, ix0 = tx * TW * d.isx + dx0;
        const long x_img = (long)b * d.hi * d.wi;
        const long y_img = (long)b * d.ho * d.wo * a.dy_pix_stride;
        __syncthreads();
        // ---- stage x' (halo tile, 64 channels) --------------------------------------------------------------------------
        if constexpr (!FIN) {
            (void)iy0; (void)ix0; (void)x_img; (void)y_img;
            for (int pix0 = 0; pix0 < npix; pix0 += XB * pstep) {
                load_x(tile, pix0);
                store_x(pix0);
            }
            load_y(tile);
            store_y();
        } else {
            const int c4 = tid & 15, p0 = tid >> 4, pstep = kThreads / 16;
            const int ch = ci0 + c4 * 4;
            const bool ch_ok = ch < d.ci;
            const bool pro = a.in_scale != nullptr;
            float sc[4], sh[4];
            if (pro) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    sc[e] = ch_ok ? a.in_scale[ch + e] : 0.0f;
                    sh[e] = ch_ok ? a.in_shift[ch + e] : 0.0f;
                }
            }
            const float* xg = (const float*)a.x + x_img * d.x_pix_stride;
            unsigned char* xs_lo = xs + a.x_plane_bytes;
            int ly = (int)(((float)p0 + 0.5f) * inv_w);
            int lx = p0 - ly * in_w;
            const int step_y = pstep / in_w, step_x = pstep - step_y * in_w;  // (uniform)
            for (int pix0 = p0; pix0 < npix; pix0 += XB * pstep) {
                float4 v[XB];
                bool ok[XB];
#pragma unroll
                for (int u = 0; u < XB; u++) {
                    const int pix = pix0 + u * pstep;
                    const int iy = iy0 + gdy + ly * row_mul, ix = ix0 + lx;
                    ok[u] = pix < npix && ch_ok && (unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi;
                    const int off = ok[u] ? (iy * d.wi + ix) * d.x_pix_stride + ch : 0;  // (a sample has < 2^31 elements)
                    lx += step_x;
                    ly += step_y;
                    if (lx >= in_w) {
                        lx -= in_w;
                        ly++;
                    }
                    v[u] = *reinterpret_cast<const float4*>(xg + off);
                }
#pragma unroll
                for (int u = 0; u < XB; u++) {
                    const int pix = pix0 + u * pstep;
                    if (pix >= npix) continue;
                    float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    unsigned hi2[2], lo2[2];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (pro) {
                            f[e] = fmaf(f[e], sc[e], sh[e]);
                            if (d.in_relu) f[e] = fmaxf(f[e], 0.0f);
                        }
                        if (!ok[u]) f[e] = 0.0f;
                    }
                    if constexpr (F32) {
                        *reinterpret_cast<float4*>(xs + pix * PSX + c4 * 16) = make_float4(f[0], f[1], f[2], f[3]);
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
                        hi2[e] = pack_bf16(h0, h1);
                        lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
                    }
                    *reinterpret_cast<uint2*>(xs + pix * PSX + c4 * 8) = make_uint2(hi2[0], hi2[1]);
                    *reinterpret_cast<uint2*>(xs_lo + pix * PSX + c4 * 8) = make_uint2(lo2[0], lo2[1]);
                }
            }
            const float* yg = (const float*)a.dy;
            const int chy = co0 + c4 * 4;
            const bool chy_ok = chy < d.co;  // (co % 4 == 0 is required in this mode)
            for (int u = 0; u < MPIX / (kThreads / 16); u++) {
                const int m = p0 + u * pstep;
                const int vy = ty * TH + (m >> 5), vx = tx * TW + (m & 31);
                const int oy = vy * d.osy + ooy, ox = vx * d.osx + oox;
                const bool ok = chy_ok && vy < d.hv && vx < d.wv && oy < d.ho && ox < d.wo;
                const int off = ok ? (oy * d.wo + ox) * a.dy_pix_stride + chy : 0;
                float4 v = *reinterpret_cast<const float4*>(yg + y_img + off);
                float f[4] = {v.x, v.y, v.z, v.w};
                unsigned hi2[2], lo2[2];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (!ok) f[e] = 0.0f;
                    if (want_bias) bsum[e] += f[e];
                }
                if constexpr (F32) {
                    *reinterpret_cast<float4*>(ys + m * PS32 + c4 * 16) = make_float4(f[0], f[1], f[2], f[3]);
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
                    hi2[e] = pack_bf16(h0, h1);
                    lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
                }
                *reinterpret_cast<uint2*>(ys + m * PSY + c4 * 8) = make_uint2(hi2[0], hi2[1]);
                *reinterpret_cast<uint2*>(ys + y_plane + m * PSY + c4 * 8) = make_uint2(lo2[0], lo2[1]);
            }
        }
        __syncthreads();
        mfma_tile();
    }
    // ---- write the slab: D[row = ci][col = co]; col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) -----------------
    const int r = lane & 31, h = lane >> 5;
    const long cip = (long)a.ci_t * CT, cop = (long)a.co_t * CT;
#pragma unroll
    for (int i = 0; i < TG; i++) {
        if (i < tcnt) {
            const int wt = d.tap_w[tb + i];
            float* base = a.slab + (((long)split * d.w_taps + wt) * cip + ci0 + ci_half * 32) * cop + co0 + co_half * 32 + r;
            // (columns beyond the real output channels are never read by the reduction: the 3-channel heads write 3 of 64 columns,
            // 24 MB of slab traffic per launch otherwise)
            if (co0 + co_half * 32 + r < (d.wgrad_co > 0 ? d.wgrad_co : d.co)) {
                const int rows_left = d.ci - (ci0 + ci_half * 32);  // (rows beyond the real input channels are never read either)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row < rows_left) base[(long)row * cop] = acc[i][e];
                }
            }
        }
    }
    if (want_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        constexpr int NCH = FIN ? 4 : 8;   // channels per thread
        constexpr int NGR = FIN ? 16 : 8;  // channel groups
#pragma unroll
        for (int e = 0; e < NCH; e++) red[tid * NCH + e] = bsum[e];
        __syncthreads();
        if (tid < CT) {
            const int grp_c = tid / NCH, e = tid % NCH;
            float s = 0.0f;
            for (int pp = 0; pp < kThreads / NGR; pp++) s += red[(pp * NGR + grp_c) * NCH + e];
            a.bias_slab[((long)cls * a.splits + split) * cop + co0 + tid] = s;
        }
    }
}

// ---- 3x3 / stride 1 / one class, bf16: ALL NINE taps per block, row-stationary fragments, loader / MFMA wave roles -----------------
// The generic kernel above gives a block 1-3 taps: every tap group stages the same halo tile and dy tile again (3.3x the algorithmic
// HBM bytes measured, profiles/r03_detector_pmc_*), reads one A fragment pair per MFMA, and its 0.35-us MFMA phase per tile cannot
// hide a ~2-us tile load (one tile of prefetch): the launch sat at a quarter of the rate the forward kernel reaches on the same
// layer.  Here a block of 8 waves (one per CU) owns a 64 x 64 channel tile of ALL 9 taps and walks TH x 32-pixel tiles:
//   * waves 4-7 LOAD: global -> registers (the tile after next is in flight while the next one is written) -> BatchNorm / ReLU
//     prologue -> LDS buffer (k + 1) & 1; waves 0-3 MULTIPLY on buffer k & 1 (9 x 16 accumulator registers each, 32 x 32 quadrant of
//     every tap): address arithmetic, the prologue's vector ALU work and the LDS stores run beside the matrix cores instead of
//     between their phases; one barrier per tile;
//   * x and dy are staged ONCE per pixel tile for all taps (halo rows shared by the three kernel rows);
//   * fragments are reused from registers: the A fragments of input row r (three horizontal shifts) serve the kernel rows
//     ky = 0, 1, 2 with the dy fragments of output rows r, r - 1, r - 2 (kept for three input rows) -- 1.06 transposing LDS read
//     pairs per MFMA instead of 2.7;
//   * LDS pixels are 128 B apart (no padding: two buffers of a TH = 8 tile fit 160 KB); the two 64-B halves of a pixel are swapped
//     where bit 1 of the pixel index is set, so the four pixels a 32-lane group of ds_read_b64_tr_b16 touches fall into four
//     different 16-bank ranges for any starting pixel (the horizontal taps start at every alignment).
// Slabs, bias sums and the fixed-order reduction are those of the generic kernel.
constexpr int kRsThreads = 512;
constexpr int RPS = 128;  // LDS bytes per pixel and plane (64 bf16 channels)

// MODE = LISO_CONV_BF16 (TH = 8 | 4) or LISO_CONV_F32X3 (fp32 tensors split into bf16 hi + lo planes while they are staged, three
// MFMAs per product; TH = 3: two buffers of two planes fit 160 KB)
// S = 1 | 2: the convolution's stride (3x3, padding 1).  S = 2: the halo tile holds (2 TH + 1) x 65 input pixels; a fragment's 16
// pixels are every second pixel of a row, so pixel pairs (2, 3), (6, 7), ... swap their LDS slots and the 64-B half swap follows bit 2
// (the four pixels of a transposing read then again fall into four different 16-bank ranges); input rows 2o and 2o + 2 of the tile
// serve kernel rows 0 and 2 of neighbouring output rows from the same fragments.
// CIW = 64 | 32 input channels per block.  32: the four MFMA waves are (output-channel half, 16-pixel column half) instead of (input half,
// output half) -- every wave multiplies one column half of every tile row, the two column halves' sums are added through LDS behind the
// pixel loop (fixed order) and a block writes 9 x 32 x 64 sums: half the slab bytes per launch and per reduction, twice the tiles per
// block.  At B = 2-4 a block has only 4 tiles of work and its slab (147 KB, written and read once more by the reduction) costs more
// memory time than its inputs: profiles/r06_wgrad_stamps.txt.
template <int MODE, int TH, int S, int CIW>
__global__ __launch_bounds__(kRsThreads, 1) void conv_wgrad_rs3_kernel(const liso_conv_desc d, const WgArgs a) {
    constexpr bool X3 = MODE == LISO_CONV_F32X3;
    constexpr int PLANES = X3 ? 2 : 1;
    constexpr int IW = (TW - 1) * S + 3, IH = (TH - 1) * S + 3;  // halo tile (padding 1)
    constexpr int NPX = IH * IW;                     // halo pixels
    constexpr int NSLOT = (NPX + 3) / 4 * 4;         // LDS pixel slots (S = 2 swaps pixel pairs: the last pair may reach NPX)
    constexpr int CPP = X3 ? 16 : 8;                 // 16-B global chunks per pixel (64 channels): dy
    constexpr int PSL = 256 / CPP;                   // pixel slots of the 256 loader threads: dy
    constexpr int CPPX = CPP * CIW / 64;             // ... of the x tile (CIW channels)
    constexpr int PSLX = 256 / CPPX;
    constexpr int XB = (NPX + PSLX - 1) / PSLX;      // chunks per loader thread
    constexpr int YB = TH * TW / PSL;                // dy: TH x 32 pixels
    constexpr int X_PLANE = NSLOT * RPS, Y_PLANE = TH * TW * RPS;
    constexpr int X_BYTES = PLANES * X_PLANE;
    constexpr int BUF = X_BYTES + PLANES * Y_PLANE;
    constexpr int NE = X3 ? 4 : 8;                   // channels per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int t = blockIdx.x;
    const int split = t % a.splits;
    t /= a.splits;
    const int cot = t % a.co_t;
    const int cit = t / a.co_t;
    const int ci0 = cit * CIW, co0 = cot * CT;
    const long cip = (long)a.ci_t * CIW, cop = (long)a.co_t * CT;
    const int n_mine = split < a.n_tiles ? (a.n_tiles - split + a.splits - 1) / a.splits : 0;  // tiles of this block

    if (wave >= 4) {
        // ================================ loader waves ================================
        const int ltid = tid - 256;
        const int cc = ltid % CPP, p0 = ltid / CPP;      // dy: chunk of the pixel, first pixel slot
        const int ccx = ltid % CPPX, p0x = ltid / CPPX;  // x
        uint4 xv[XB], yv[YB];
        unsigned xok = 0u, yok = 0u;
        float sc[NE], sh[NE];
        float bsum[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) bsum[e] = 0.0f;
        const bool want_bias = a.bias_slab != nullptr && cit == 0;
        const bool pro = a.in_scale != nullptr;
        const int ch = ci0 + ccx * NE, chy = co0 + cc * NE;
        const bool ch_ok = ch < d.ci, chy_ok = chy < d.co;
        if (pro) {
#pragma unroll
            for (int e = 0; e < NE; e++) {
                sc[e] = ch_ok ? a.in_scale[ch + e] : 0.0f;
                sh[e] = ch_ok ? a.in_shift[ch + e] : 0.0f;
            }
        }
        // this thread's halo pixels: position inside the tile, element offset relative to the tile's first pixel
        int rel[XB];
        unsigned lyx[XB];
#pragma unroll
        for (int u = 0; u < XB; u++) {
            const int pix = p0x + u * PSLX;
            const int ly = pix / IW, lx = pix - ly * IW;
            rel[u] = (ly * d.wi + lx) * d.x_pix_stride;
            lyx[u] = (unsigned)ly | ((unsigned)lx << 8) | (pix < NPX && ch_ok ? 0x10000u : 0u);
        }
        auto load_tile = [&](int tile) {
            const int tx = tile % a.tiles_x;
            const int tq = tile / a.tiles_x;
            const int ty = tq % a.tiles_y, b = tq / a.tiles_y;
            const int iy0 = ty * TH * S - 1, ix0 = tx * TW * S - 1;
            const long xbase = (long)b * d.hi * d.wi * d.x_pix_stride + ch;
            const int org = (iy0 * d.wi + ix0) * d.x_pix_stride;
            const bool inner = iy0 >= 0 && iy0 + IH <= d.hi && ix0 >= 0 && ix0 + IW <= d.wi;  // (uniform) no pixel outside the image
            xok = 0u;
#pragma unroll
            for (int u = 0; u < XB; u++) {
                bool ok = (lyx[u] >> 16) != 0u;
                if (!inner) {
                    const int iy = iy0 + (int)(lyx[u] & 0xffu), ix = ix0 + (int)((lyx[u] >> 8) & 0xffu);
                    ok = ok && (unsigned)iy < (unsigned)d.hi && (unsigned)ix < (unsigned)d.wi;
                }
                xok |= ok ? (1u << u) : 0u;
                const long off = xbase + (ok ? org + rel[u] : 0);
                if constexpr (X3)
                    xv[u] = *reinterpret_cast<const uint4*>((const float*)a.x + off);
                else
                    xv[u] = *reinterpret_cast<const uint4*>((const unsigned short*)a.x + off);
            }
            const long ybase = (long)b * d.ho * d.wo * a.dy_pix_stride + chy;
            yok = 0u;
#pragma unroll
            for (int u = 0; u < YB; u++) {
                const int m = p0 + u * PSL;
                const int oy = ty * TH + (m >> 5), ox = tx * TW + (m & 31);
                const bool ok = chy_ok && oy < d.ho && ox < d.wo;
                yok |= ok ? (1u << u) : 0u;
                const long off = ybase + (ok ? (long)(oy * d.wo + ox) * a.dy_pix_stride : 0);
                if constexpr (X3)
                    yv[u] = *reinterpret_cast<const uint4*>((const float*)a.dy + off);
                else
                    yv[u] = *reinterpret_cast<const uint4*>((const unsigned short*)a.dy + off);
            }
        };
        auto store_tile = [&](unsigned char* buf) {
            unsigned char* xs = buf;
            unsigned char* ys = buf + X_BYTES;
#pragma unroll
            for (int u = 0; u < XB; u++) {
                const int pix = p0x + u * PSLX;
                if (pix >= NPX) continue;
                const int swz = S == 1 ? ((pix >> 1) & 1) << 6 : ((pix >> 2) & 1) << 6;
                const int slot = S == 1 ? pix : pix ^ ((pix >> 1) & 1);
                if constexpr (X3) {
                    float f[4] = {__uint_as_float(xv[u].x), __uint_as_float(xv[u].y), __uint_as_float(xv[u].z), __uint_as_float(xv[u].w)};
                    unsigned hi2[2], lo2[2];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (pro) {
                            f[e] = fmaf(f[e], sc[e], sh[e]);
                            if (d.in_relu) f[e] = fmaxf(f[e], 0.0f);
                        }
                        if (!((xok >> u) & 1u)) f[e] = 0.0f;
                    }
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
                        hi2[e] = pack_bf16(h0, h1);
                        lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
                    }
                    *reinterpret_cast<uint2*>(xs + slot * RPS + ((ccx * 8) ^ swz)) = make_uint2(hi2[0], hi2[1]);
                    *reinterpret_cast<uint2*>(xs + X_PLANE + slot * RPS + ((ccx * 8) ^ swz)) = make_uint2(lo2[0], lo2[1]);
                } else {
                    uint4 o = xv[u];
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
                    if (!((xok >> u) & 1u)) o = make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(xs + slot * RPS + ((ccx * 16) ^ swz)) = o;
                }
            }
#pragma unroll
            for (int u = 0; u < YB; u++) {
                const int m = p0 + u * PSL;
                const int swz = ((m >> 1) & 1) << 6;
                uint4 v = yv[u];
                if (!((yok >> u) & 1u)) v = make_uint4(0u, 0u, 0u, 0u);
                if constexpr (X3) {
                    const float f[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
                    unsigned hi2[2], lo2[2];
#pragma unroll
                    for (int e = 0; e < 2; e++) {
                        const float h0 = round_bf16(f[2 * e]), h1 = round_bf16(f[2 * e + 1]);
                        hi2[e] = pack_bf16(h0, h1);
                        lo2[e] = pack_bf16(f[2 * e] - h0, f[2 * e + 1] - h1);
                    }
                    *reinterpret_cast<uint2*>(ys + m * RPS + ((cc * 8) ^ swz)) = make_uint2(hi2[0], hi2[1]);
                    *reinterpret_cast<uint2*>(ys + Y_PLANE + m * RPS + ((cc * 8) ^ swz)) = make_uint2(lo2[0], lo2[1]);
                    if (want_bias) {
#pragma unroll
                        for (int e = 0; e < 4; e++) bsum[e] += f[e];
                    }
                } else {
                    *reinterpret_cast<uint4*>(ys + m * RPS + ((cc * 16) ^ swz)) = v;
                    if (want_bias) {
                        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            bsum[2 * e] += bf16_lo(w[e]);
                            bsum[2 * e + 1] += bf16_hi(w[e]);
                        }
                    }
                }
            }
        };
#ifdef LISO_WGRAD_STAMPS
        unsigned long long ws_store = 0, ws_load = 0, ws_bar = 0, ws_t;
        const unsigned long long ws_k0 = __builtin_amdgcn_s_memtime();
#define WSTAMP_BEGIN ws_t = __builtin_amdgcn_s_memtime();
#define WSTAMP_END(acc) { const unsigned long long ws_n = __builtin_amdgcn_s_memtime(); acc += ws_n - ws_t; ws_t = ws_n; }
#else
#define WSTAMP_BEGIN
#define WSTAMP_END(acc)
#endif
        if (n_mine > 0) {
            load_tile(split);
            store_tile(smem);
            if (n_mine > 1) load_tile(split + a.splits);
        }
        __syncthreads();  // (A) buffer 0 is ready
#ifdef LISO_WGRAD_STAMPS
        const unsigned long long ws_k1 = __builtin_amdgcn_s_memtime();
#endif
        for (int k = 0; k < n_mine; k++) {
            WSTAMP_BEGIN
            if (k + 1 < n_mine) {
                store_tile(smem + ((k + 1) & 1) * BUF);  // the tile the MFMA waves take next (they left this buffer at barrier k - 1)
                WSTAMP_END(ws_store)
                if (k + 2 < n_mine) load_tile(split + (k + 2) * a.splits);
                WSTAMP_END(ws_load)
            }
            __syncthreads();  // (B k)
            WSTAMP_END(ws_bar)
        }
#ifdef LISO_WGRAD_STAMPS
        if (ltid == 0 && a.stamps) {
            unsigned long long* o = a.stamps + (size_t)blockIdx.x * 16 + 8;
            o[0] = ws_k1 - ws_k0; o[1] = ws_store; o[2] = ws_load; o[3] = ws_bar; o[4] = (unsigned long long)n_mine;
        }
#endif
        if (want_bias) {  // (the MFMA waves are past their last LDS read: barrier B of the last tile)
            float* red = reinterpret_cast<float*>(smem);
#pragma unroll
            for (int e = 0; e < NE; e++) red[ltid * NE + e] = bsum[e];
        }
        __syncthreads();  // (C)
        if (want_bias && ltid < CT) {
            const float* red = reinterpret_cast<const float*>(smem);
            const int grp_c = ltid / NE, e = ltid % NE;
            float s_ = 0.0f;
            for (int pp = 0; pp < PSL; pp++) s_ += red[(pp * CPP + grp_c) * NE + e];
            a.bias_slab[(long)split * cop + co0 + ltid] = s_;
        }
        if constexpr (CIW == 32) __syncthreads();  // (D) the MFMA waves' hand-over below
        return;
    }
    // ================================ MFMA waves ================================
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // CIW = 64: wave = (input-channel half, output-channel half), both 16-pixel column halves of a tile row.
    // CIW = 32: wave = (column half, output-channel half), the block's 32 input channels.
    const int ci_half = CIW == 64 ? wave >> 1 : 0, co_half = wave & 1;
    const int c_lo = CIW == 64 ? 0 : wave >> 1, c_hi = CIW == 64 ? 2 : (wave >> 1) + 1;
    // lane offsets of the transposing reads, relative to the fragment's first pixel `base`, for every alignment of `base` that changes
    // them.  S = 1: pixel base + 8 (g >> 1) + q, the half swap follows bit 1 (4 alignments; + 4 or + 8 pixels never change it).
    // S = 2: pixel base + 2 (8 (g >> 1) + q), slot = pixel ^ bit 1, half swap on bit 2 (8 alignments; + 8 or + 16 change neither).
    constexpr int NAL = S == 1 ? 4 : 8;
    int a_lane[NAL];
#pragma unroll
    for (int m = 0; m < NAL; m++) {
        const int off = S * (8 * (g >> 1) + q), pm = m + S * q;  // (pixel offset; the pixel's low bits)
        const int slot_off = S == 1 ? off : (off + ((pm ^ ((pm >> 1) & 1)) - pm));
        const int swz = S == 1 ? ((pm >> 1) & 1) << 6 : ((pm >> 2) & 1) << 6;
        a_lane[m] = slot_off * RPS + ((ci_half * 64 + 32 * (g & 1) + 8 * p) ^ swz);
    }
    const int b_lane = (8 * (g >> 1) + q) * RPS + ((co_half * 64 + 32 * (g & 1) + 8 * p) ^ (((q >> 1) & 1) << 6));

    f16v acc[9];
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;

    auto mma = [&](int i, const bf8& ah, const bf8& al, const bf8& bh, const bf8& bl) {
        if constexpr (X3) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
    };

#ifdef LISO_WGRAD_STAMPS
    unsigned long long ms_mul = 0, ms_bar = 0, ms_t;
    const unsigned long long ms_k0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();  // (A)
#ifdef LISO_WGRAD_STAMPS
    const unsigned long long ms_k1 = __builtin_amdgcn_s_memtime();
    ms_t = ms_k1;
#endif
    for (int k = 0; k < n_mine; k++) {
        const unsigned char* xs = smem + (k & 1) * BUF;
        const unsigned char* ys = xs + X_BYTES;
#pragma unroll
        for (int c = c_lo; c < c_hi; c++) {
            bf8 bf[3], bl[3];
#pragma unroll
            for (int r = 0; r < IH; r++) {
                bf8 af[3], al[3];
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int base = r * IW + 16 * S * c + kx;
                    const unsigned char* ap = xs + base * RPS + a_lane[base & (NAL - 1)];
                    af[kx] = tr_pair(ap, ap + 4 * S * RPS);
                    if constexpr (X3) al[kx] = tr_pair(ap + X_PLANE, ap + X_PLANE + 4 * S * RPS);
                }
                // the output row whose fragments enter the registers with this input row: S = 1: row r; S = 2: row r / 2 at even r
                if ((S == 1 && r < TH) || (S == 2 && (r & 1) == 0 && r / 2 < TH)) {
                    const int o = r / S;
                    const unsigned char* bp = ys + (o * TW + 16 * c) * RPS + b_lane;
                    bf[o % 3] = tr_pair(bp, bp + 4 * RPS);
                    if constexpr (X3) bl[o % 3] = tr_pair(bp + Y_PLANE, bp + Y_PLANE + 4 * RPS);
                }
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    // input row r is row ky of output row o when r = S o + ky
                    if ((r - ky) % S != 0) continue;
                    const int o = (r - ky) / S;
                    if (r - ky >= 0 && o < TH) {
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) mma(ky * 3 + kx, af[kx], al[kx], bf[o % 3], bl[o % 3]);
                    }
                }
            }
        }
#ifdef LISO_WGRAD_STAMPS
        { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ms_mul += n_ - ms_t; ms_t = n_; }
#endif
        __syncthreads();  // (B k)
#ifdef LISO_WGRAD_STAMPS
        { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ms_bar += n_ - ms_t; ms_t = n_; }
#endif
    }
    __syncthreads();  // (C)
#ifdef LISO_WGRAD_STAMPS
    const unsigned long long ms_k2 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (CIW == 32) {
        // the two column halves' sums: waves 2 | 3 hand theirs over through LDS ([wave][tap][register][lane]: 256-B rows), waves 0 | 1 add
        // them to their own (own + other: a fixed order) and write the slab.  (The first 8 KB hold the loader waves' bias rows.)
        float* hand = reinterpret_cast<float*>(smem + 8192) + (size_t)(wave & 1) * 9 * 16 * 64;
        if (wave >= 2) {
#pragma unroll
            for (int i = 0; i < 9; i++)
#pragma unroll
                for (int e = 0; e < 16; e++) hand[(i * 16 + e) * 64 + lane] = acc[i][e];
        }
        __syncthreads();  // (D)
        if (wave >= 2) return;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][e] += hand[(i * 16 + e) * 64 + lane];
    }
    // ---- slab: D[row = ci][col = co]; col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ---------------------------------
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int wt = d.tap_w[i];  // (taps are listed ky-major, kx-minor: checked by the plan)
        float* base = a.slab + (((long)split * d.w_taps + wt) * cip + ci0 + ci_half * 32) * cop + co0 + co_half * 32 + r;
        if (co0 + co_half * 32 + r < (d.wgrad_co > 0 ? d.wgrad_co : d.co)) {
            const int rows_left = d.ci - (ci0 + ci_half * 32);
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < rows_left) base[(long)row * cop] = acc[i][e];
            }
        }
    }
#ifdef LISO_WGRAD_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the slab stores have left
    if (tid == 0 && a.stamps) {
        unsigned long long* o = a.stamps + (size_t)blockIdx.x * 16;
        const unsigned long long ms_k3 = __builtin_amdgcn_s_memtime();
        o[0] = ms_k1 - ms_k0; o[1] = ms_mul; o[2] = ms_bar; o[3] = ms_k2 - ms_k0; o[4] = ms_k3 - ms_k2; o[5] = ms_k3 - ms_k0;
    }
#endif
}

// dw (torch layout) = sum over splits of the slabs, in a fixed order.  PARTS = 16: block = one (tap, k) row x 64 output channels;
// thread = 4 consecutive channels (one 16-B load per split) x one of 16 split groups (group g adds splits g, g + 16, ... in that
// order, four loads in flight), then the 16 group sums are added pairwise in a fixed tree.  PARTS = 1 (<= 16 splits): block = 16
// rows x 64 channels, every thread walks all splits of its 4 channels (all loads in flight), no tree.  The slabs were written a
// moment ago and sit in L2 / the Infinity Cache; what the reduction needs is bytes in flight (the former version: one 4-B load at a
// time per thread, 2 TB/s on 37 MB of slabs).
template <int PARTS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias_slab,
                                                           int splits, int bias_rows, int taps, int ci, int co, long cip, long cop,
                                                           int transposed, float* __restrict__ dw, float* __restrict__ dbias) {
    constexpr int RPB = 16 / PARTS;  // rows per block
    __shared__ float4 red[16][16];
    const int c4 = threadIdx.x & 15, part = PARTS == 16 ? threadIdx.x >> 4 : 0, rib = PARTS == 16 ? 0 : threadIdx.x >> 4;
    const int n_tiles = (co + 63) / 64;
    const long rows = (long)taps * ci;
    const long row_blocks = (rows + RPB - 1) / RPB;
    const long bid = blockIdx.x;
    const bool is_bias = bid >= row_blocks * n_tiles;
    if (is_bias && !dbias) return;
    const int ntile = (int)(is_bias ? bid - row_blocks * n_tiles : bid % n_tiles);
    const long row = is_bias ? 0 : (bid / n_tiles) * RPB + rib;  // tap * ci + k
    const bool row_ok = is_bias ? rib == 0 : row < rows;
    const int k = (int)(row % ci), tap = (int)(row / ci);
    const int n = ntile * 64 + c4 * 4;
    const int count = is_bias ? bias_rows : splits;
    const float* src = is_bias ? bias_slab + n : slab + ((long)tap * cip + k) * cop + n;
    const long stride = is_bias ? cop : (long)taps * cip * cop;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (n < co && row_ok) {
        int sp = part;
        for (; sp + 3 * PARTS < count; sp += 4 * PARTS) {  // four loads in flight per thread
            const float4 v0 = *reinterpret_cast<const float4*>(src + (long)sp * stride);
            const float4 v1 = *reinterpret_cast<const float4*>(src + (long)(sp + PARTS) * stride);
            const float4 v2 = *reinterpret_cast<const float4*>(src + (long)(sp + 2 * PARTS) * stride);
            const float4 v3 = *reinterpret_cast<const float4*>(src + (long)(sp + 3 * PARTS) * stride);
            s.x = (((s.x + v0.x) + v1.x) + v2.x) + v3.x;
            s.y = (((s.y + v0.y) + v1.y) + v2.y) + v3.y;
            s.z = (((s.z + v0.z) + v1.z) + v2.z) + v3.z;
            s.w = (((s.w + v0.w) + v1.w) + v2.w) + v3.w;
        }
        for (; sp < count; sp += PARTS) {
            const float4 v = *reinterpret_cast<const float4*>(src + (long)sp * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    if constexpr (PARTS == 16) {
        red[part][c4] = s;
        __syncthreads();
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1) {  // fixed pairing: (p, p + w)
            if (part < w) {
                const float4 o = red[part + w][c4];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
                red[part][c4] = s;
            }
            __syncthreads();
        }
    }
    if (part == 0 && n < co && row_ok) {
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (n + e >= co) break;
            if (is_bias)
                dbias[n + e] = v[e];
            else
                dw[transposed ? (((long)k * co + n + e) * taps + tap) : (((long)(n + e) * ci + k) * taps + tap)] = v[e];
        }
    }
}

int launch_reduce(const float* slab, const float* bias_slab, int splits, int bias_rows, int taps, int ci, int co_w, long cip, long cop,
                  int transposed, float* dw, float* dbias, hipStream_t st) {
    const long rows = (long)taps * ci;
    const int n_tiles = (co_w + 63) / 64;
    if (splits > 16) {
        const long blocks = (rows + (dbias ? 1 : 0)) * n_tiles;
        wgrad_reduce_kernel<16><<<(int)blocks, 256, 0, st>>>(slab, bias_slab, splits, bias_rows, taps, ci, co_w, cip, cop, transposed, dw, dbias);
    } else {
        const long blocks = ((rows + 15) / 16 + (dbias ? 1 : 0)) * n_tiles;
        wgrad_reduce_kernel<1><<<(int)blocks, 256, 0, st>>>(slab, bias_slab, splits, bias_rows, taps, ci, co_w, cip, cop, transposed, dw, dbias);
    }
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct WgPlan {
    int tg, lds, splits, blocks;
    size_t slab_bytes, bias_bytes;
    WgArgs a;
};

bool make_plan_impl(const liso_conv_desc& d, WgPlan* p, bool compact) {
    if (d.batch <= 0 || d.ci <= 0 || d.co <= 0 || d.n_classes < 1 || d.n_classes > LISO_CONV_MAX_CLASSES) return false;
    if (d.n_taps < 1 || d.n_taps > LISO_CONV_MAX_TAPS || d.class_tap_begin[0] != 0 || d.class_tap_begin[d.n_classes] != d.n_taps)
        return false;
    if (d.mode != LISO_CONV_BF16 && d.mode != LISO_CONV_F32X3 && d.mode != LISO_CONV_F32) return false;
    const bool f32 = d.mode == LISO_CONV_F32;
    const bool x3 = d.mode == LISO_CONV_F32X3 || f32;  // (plan: the exact mode shares the fp32-tensor choices of F32X3)
    const int vec = x3 ? 4 : 8;
    if (d.ci % vec || d.co % vec || d.x_pix_stride % vec) return false;
    const int planes = (x3 && !f32) ? 2 : 1;
    const int psx = f32 ? PS32 : PSB, psy = f32 ? PS32 : PSB;
    WgArgs& a = p->a;
    a.ci_t = (d.ci + CT - 1) / CT;
    a.co_t = (d.co + CT - 1) / CT;
    a.tiles_x = (d.wv + TW - 1) / TW;
    int max_cls_taps = 0, TH = 4;
    // tile rows: the tallest tile whose LDS image allows 2 blocks per CU (79 KB), else the tallest that fits at all
    int th_fit = 0;
    for (int pass = 0; pass < 2 && !th_fit; pass++)
        for (int th = 4; th >= 1 && !th_fit; th >>= 1) {
            int mp = 0;
            for (int c = 0; c < d.n_classes; c++) {
                int y0 = 1 << 30, y1 = -(1 << 30), x0 = 1 << 30, x1 = -(1 << 30);
                for (int t = d.class_tap_begin[c]; t < d.class_tap_begin[c + 1]; t++) {
                    y0 = d.tap_dy[t] < y0 ? d.tap_dy[t] : y0;
                    y1 = d.tap_dy[t] > y1 ? d.tap_dy[t] : y1;
                    x0 = d.tap_dx[t] < x0 ? d.tap_dx[t] : x0;
                    x1 = d.tap_dx[t] > x1 ? d.tap_dx[t] : x1;
                }
                const int np = (compact ? th : (th - 1) * d.isy + (y1 - y0) + 1) * ((TW - 1) * d.isx + (x1 - x0) + 1);
                mp = np > mp ? np : mp;
            }
            const int lds = planes * (round_up(mp * psx, 16) + th * TW * psy);
            if (lds <= (pass == 0 ? 79 : 158) * 1024) th_fit = th;
        }
    if (!th_fit) return false;
    TH = th_fit;
    a.th = TH;
    a.tiles_y = (d.hv + TH - 1) / TH;
    a.n_tiles = d.batch * a.tiles_y * a.tiles_x;
    int max_pix = 0;
    for (int c = 0; c < d.n_classes; c++) {
        int y0 = 1 << 30, y1 = -(1 << 30), x0 = 1 << 30, x1 = -(1 << 30);
        const int nt = d.class_tap_begin[c + 1] - d.class_tap_begin[c];
        if (nt < 1) return false;
        max_cls_taps = nt > max_cls_taps ? nt : max_cls_taps;
        for (int t = d.class_tap_begin[c]; t < d.class_tap_begin[c + 1]; t++) {
            y0 = d.tap_dy[t] < y0 ? d.tap_dy[t] : y0;
            y1 = d.tap_dy[t] > y1 ? d.tap_dy[t] : y1;
            x0 = d.tap_dx[t] < x0 ? d.tap_dx[t] : x0;
            x1 = d.tap_dx[t] > x1 ? d.tap_dx[t] : x1;
            if (d.tap_w[t] < 0 || d.tap_w[t] >= d.w_taps) return false;
        }
        a.cls_dy0[c] = y0;
        a.cls_dx0[c] = x0;
        a.cls_inh[c] = compact ? TH : (TH - 1) * d.isy + (y1 - y0) + 1;
        a.cls_inw[c] = (TW - 1) * d.isx + (x1 - x0) + 1;
        const int np = a.cls_inh[c] * a.cls_inw[c];
        max_pix = np > max_pix ? np : max_pix;
    }
    a.psx = psx;
    a.compact = compact ? 1 : 0;
    // (XB = 8 chunks per thread x 32 pixel rows of threads = 256 halo pixels; XB = 4 in the 9-tap instantiation)
    a.pipelined = 0;
    a.x_plane_bytes = round_up(max_pix * a.psx, 16);
    p->lds = planes * (a.x_plane_bytes + TH * TW * psy);
    // taps per block (9 / 3 / 1) and pixel splits: the slabs of all splits together stay below 24 MB (they are written and
    // read once), every block sees >= 4 tiles, and the grid should reach ~2 blocks per CU; more taps per block = fewer
    // re-stagings of the same tiles, so the largest tap group that still fills the chip wins.
    const long cc = (long)a.ci_t * a.co_t;
    auto groups = [&](int tg) {
        int n = 0;
        for (int c = 0; c < d.n_classes; c++) n += (d.class_tap_begin[c + 1] - d.class_tap_begin[c] + tg - 1) / tg;
        return n;
    };
    const long slab_per_split = (long)d.w_taps * a.ci_t * CT * a.co_t * CT * sizeof(float);
    long slab_mb = 24;
    if (const char* e = getenv("LISO_WGRAD_SLAB_MB")) slab_mb = atol(e) > 0 ? atol(e) : slab_mb;  // experiments
    long split_cap = (slab_mb << 20) / slab_per_split;
    // tiles per block at least (the first tile's load is exposed).  fp32 tensors: 2 -- the SLIM encoders' 1x1 / strided layers have 256
    // tiles in all: at 4 per block 64 of the 256 CUs worked, each through four unpipelined load -> multiply rounds (28 us per launch);
    // measured on the SLIM train step: weight-gradient family 2.10-2.13 -> 2.00-2.05 ms (1 tile per block: no better, 4x the slabs)
    long min_tiles = x3 ? 2 : 4;
    if (const char* e = getenv("LISO_WGRAD_MIN_TILES")) min_tiles = atol(e) > 0 ? atol(e) : min_tiles;  // experiments
    const long by_tiles = a.n_tiles >= min_tiles ? a.n_tiles / min_tiles : 1;
    split_cap = split_cap < 1 ? 1 : (split_cap > by_tiles ? by_tiles : split_cap);
    long target = 512;  // ~2 blocks per CU
    if (const char* e = getenv("LISO_WGRAD_BLOCKS")) target = atol(e) > 0 ? atol(e) : target;  // experiments
    // 7 x 7 kernels (the SLIM encoders' stem, the motion encoder's flow / class convolutions): one kernel ROW of 7 taps per block
    const bool rows7 = max_cls_taps == 49 && d.n_classes == 1;
    const int tg_opts[3] = {rows7 ? 7 : 9, rows7 ? 1 : 3, 1};
    long best_blocks = -1;
    p->tg = 1;
    for (int k = ((!rows7 && (x3 || compact)) ? 1 : 0); k < 3; k++) {  // (compact tiles hold one kernel row: no 9-tap groups)
        const int tg = tg_opts[k];
        if (tg > 1 && max_cls_taps == 1) continue;
        const long per_split = cc * groups(tg);
        long s = (target + per_split - 1) / per_split;
        s = s < 1 ? 1 : (s > split_cap ? split_cap : s);
        const long blocks = per_split * s;
        if (blocks >= 256 || (rows7 && tg == 7)) {  // enough: take the largest tap group (7 x 7: always whole kernel rows -- single
                                                     // taps would stage the halo tile 49 times)
            p->tg = tg;
            break;
        }
        if (blocks > best_blocks) {
            best_blocks = blocks;
            p->tg = tg;
        }
    }
    if (const char* e = getenv("LISO_WGRAD_TG")) {  // experiments: force the tap group (9 / 3 / 1)
        const int v = atoi(e);
        if ((v == 9 && !x3 && !compact && !rows7) || (v == 3 && !rows7) || v == 1 || (v == 7 && rows7)) p->tg = v;
    }
    {
        const int xb = p->tg >= 9 ? 4 : 8;
        a.pipelined = (!x3 && max_pix <= xb * (kThreads / 8)) ? 1 : 0;
        if (const char* e = getenv("LISO_WGRAD_PIPE")) a.pipelined = a.pipelined && atoi(e) != 0;  // experiments
    }
    a.n_groups = 0;
    for (int c = 0; c < d.n_classes; c++)
        for (int t = d.class_tap_begin[c]; t < d.class_tap_begin[c + 1]; t += p->tg) {
            if (a.n_groups >= LISO_CONV_MAX_TAPS) return false;
            a.grp_cls[a.n_groups] = c;
            a.grp_begin[a.n_groups] = t;
            const int left = d.class_tap_begin[c + 1] - t;
            a.grp_cnt[a.n_groups] = left < p->tg ? left : p->tg;
            if (compact)  // every tap of the group must sit in the same kernel row
                for (int q = 1; q < a.grp_cnt[a.n_groups]; q++)
                    if (d.tap_dy[t + q] != d.tap_dy[t]) return false;
            a.n_groups++;
        }
    const long per_split = cc * a.n_groups;
    long s = (target + per_split - 1) / per_split;
    s = s < 1 ? 1 : (s > split_cap ? split_cap : s);
    p->splits = (int)s;
    a.splits = p->splits;
    p->blocks = (int)(per_split * s);
    p->slab_bytes = (size_t)s * d.w_taps * a.ci_t * CT * a.co_t * CT * sizeof(float);
    p->bias_bytes = (size_t)d.n_classes * s * a.co_t * CT * sizeof(float);
    return true;
}

// Strided convolutions (the backbone's three stride-2 layers): a tap group of one kernel row reads every isy-th input row only --
// its tile then holds those rows alone (2.5x fewer staged pixels for 3x3 / 2, and twice the output rows per tile in the same LDS).
bool make_plan(const liso_conv_desc& d, WgPlan* p) {
    bool want = d.isy > 1;
    if (const char* e = getenv("LISO_WGRAD_COMPACT")) want = want && atoi(e) != 0;  // experiments
    if (want && make_plan_impl(d, p, true)) return true;
    return make_plan_impl(d, p, false);
}

template <int MODE, int TG>
int launch(const liso_conv_desc& d, const WgPlan& p, hipStream_t st) {
    static liso_dev::PerDeviceFlag attr_set;
    if (!liso_dev::lds_opt_in(attr_set, (const void*)conv_wgrad_kernel<MODE, TG>, 160 * 1024)) return LISO_ELAUNCH;
    conv_wgrad_kernel<MODE, TG><<<p.blocks, kThreads, p.lds, st>>>(d, p.a);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

// ---- sparse-input weight gradient (the encoders' 7x7 / 2 stem on the pillar canvas) ----------------------------------------------
// The canvas holds a feature row at 1-2 % of its cells (pillar_scatter.py:62-102); every other cell is exactly zero and contributes
// nothing to dW[tap][ci][co] = sum_cells x[cell][ci] * dy[out pixel(cell, tap)][co].  The dense kernels stage the whole 134-MB fp32
// canvas (once per kernel row), the library's kernel takes 0.31 ms per two sweeps; here
//   cells_count / cells_scan / cells_fill   list the occupied cells in (sample, row, column) order from the occupancy map
//                                           (per-row ballot counts, one-block scan of the row counts: deterministic order);
//   wgrad_sparse_kernel                     block = (tap, split): walks its share of the list, keeps the cells whose parity puts an
//                                           output pixel under this tap (1 of 4 at stride 2), stages their feature rows and dy rows
//                                           in LDS and accumulates the 64 x 64 tile with fp32 FMAs (exact fp32 products: the
//                                           arithmetic of LISO_CONV_F32, better than F32X3), 8 outputs per thread;
//   dy_colsum_kernel                        the bias gradient (a dense column sum of dy) as rows of the bias slab;
// and the fixed-order slab reduction of the dense kernels.  Everything is order-fixed: bitwise reproducible.
constexpr int kSpSplits = 16;
constexpr int kSpBatch = 64;  // candidate cells per staging round

__global__ __launch_bounds__(256) void cells_count_kernel(const float* __restrict__ occ, int wi, int* __restrict__ row_count) {
    __shared__ int wsum[4];
    const long row = blockIdx.x;
    int n = 0;
    for (int c = threadIdx.x; c < wi; c += 256) n += occ[row * wi + c] != 0.0f ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) row_count[row] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one block: row_off[r] = exclusive prefix of row_count, row_off[n_rows] = total
__global__ __launch_bounds__(1024) void cells_scan_kernel(const int* __restrict__ row_count, int n_rows, int* __restrict__ row_off) {
    __shared__ int part[1024];
    const int per = (n_rows + 1023) / 1024;
    const int b0 = threadIdx.x * per;
    int s = 0;
    for (int i = 0; i < per; i++)
        if (b0 + i < n_rows) s += row_count[b0 + i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x > 0 ? part[threadIdx.x - 1] : 0;
    for (int i = 0; i < per; i++)
        if (b0 + i < n_rows) {
            row_off[b0 + i] = run;
            run += row_count[b0 + i];
        }
    if (threadIdx.x == 1023) row_off[n_rows] = part[1023];
}

__global__ __launch_bounds__(64) void cells_fill_kernel(const float* __restrict__ occ, int wi, const int* __restrict__ row_off,
                                                        int* __restrict__ cells) {
    const long row = blockIdx.x;
    int base = row_off[row];
    for (int c0 = 0; c0 < wi; c0 += 64) {
        const int c = c0 + threadIdx.x;
        const bool on = c < wi && occ[row * wi + c] != 0.0f;
        const unsigned long long m = __ballot(on);
        if (on) cells[base + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = (int)(row * wi + c);
        base += __popcll(m);
    }
}

struct SpArgs {
    const float* x;
    const float* dy;
    const int* cells;
    const int* n_cells;  // row_off[n_rows]
    float* slab;         // [splits][w_taps][64][64]
    int dy_pix_stride;
    int pad;             // -tap_dy of kernel row 0 (= the convolution's padding)
};

__global__ __launch_bounds__(256) void wgrad_sparse_kernel(const liso_conv_desc d, const SpArgs a) {
    __shared__ __attribute__((aligned(16))) float xs[kSpBatch][64];
    __shared__ __attribute__((aligned(16))) float ys[kSpBatch][64];
    __shared__ int src_x[kSpBatch], src_y[kSpBatch], n_sel;
    const int tap = blockIdx.x, split = blockIdx.y, tid = threadIdx.x;
    const int tdy = d.tap_dy[tap], tdx = d.tap_dx[tap];
    const int n = *a.n_cells;
    const int chunk = (n + kSpSplits - 1) / kSpSplits;
    const int begin = split * chunk, end = begin + chunk < n ? begin + chunk : n;
    const int ci = tid >> 2, co8 = (tid & 3) * 16;  // thread: one input channel x 16 output channels (two 8-groups: co8, co8 + 8)
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    const int plane = d.hi * d.wi;
    for (int c0 = begin; c0 < end; c0 += kSpBatch) {
        __syncthreads();  // (the previous round's LDS rows are done)
        if (tid < 64) {   // wave 0 selects: input cell (r, c) lies under tap (tdy, tdx) of output pixel (oy, ox) iff r = oy * isy + tdy
            bool ok = false;
            int sx = 0, sy = 0;
            if (c0 + tid < end) {
                const int cell = a.cells[c0 + tid];
                const int b = cell / plane, rc = cell - b * plane;
                const int r = rc / d.wi, c = rc - r * d.wi;
                const int ny = r - tdy, nx = c - tdx;
                if (ny >= 0 && nx >= 0 && ny % d.isy == 0 && nx % d.isx == 0) {
                    const int oy = ny / d.isy, ox = nx / d.isx;
                    if (oy < d.ho && ox < d.wo) {
                        ok = true;
                        sx = cell;
                        sy = (b * d.ho + oy) * d.wo + ox;
                    }
                }
            }
            const unsigned long long m = __ballot(ok);
            if (ok) {
                const int k = __popcll(m & ((1ull << tid) - 1ull));
                src_x[k] = sx;
                src_y[k] = sy;
            }
            if (tid == 0) n_sel = __popcll(m);
        }
        __syncthreads();
        const int ns = n_sel;
        // stage the selected rows: 16 float4 per x row (64 channels, zero beyond ci) and per dy row (zero beyond co)
        for (int i = tid; i < ns * 16; i += 256) {
            const int j = i >> 4, q4 = (i & 15) * 4;
            float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vy = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q4 < d.ci) vx = *reinterpret_cast<const float4*>(a.x + (long)src_x[j] * d.x_pix_stride + q4);
            if (q4 < d.co) vy = *reinterpret_cast<const float4*>(a.dy + (long)src_y[j] * a.dy_pix_stride + q4);
            *reinterpret_cast<float4*>(&xs[j][q4]) = vx;
            *reinterpret_cast<float4*>(&ys[j][q4]) = vy;
        }
        __syncthreads();
        for (int j = 0; j < ns; j++) {
            const float xv = xs[j][ci];
            const float4 y0 = *reinterpret_cast<const float4*>(&ys[j][co8]), y1 = *reinterpret_cast<const float4*>(&ys[j][co8 + 4]);
            const float4 y2 = *reinterpret_cast<const float4*>(&ys[j][co8 + 8]), y3 = *reinterpret_cast<const float4*>(&ys[j][co8 + 12]);
            acc[0] = fmaf(xv, y0.x, acc[0]); acc[1] = fmaf(xv, y0.y, acc[1]); acc[2] = fmaf(xv, y0.z, acc[2]); acc[3] = fmaf(xv, y0.w, acc[3]);
            acc[4] = fmaf(xv, y1.x, acc[4]); acc[5] = fmaf(xv, y1.y, acc[5]); acc[6] = fmaf(xv, y1.z, acc[6]); acc[7] = fmaf(xv, y1.w, acc[7]);
            acc[8] = fmaf(xv, y2.x, acc[8]); acc[9] = fmaf(xv, y2.y, acc[9]); acc[10] = fmaf(xv, y2.z, acc[10]); acc[11] = fmaf(xv, y2.w, acc[11]);
            acc[12] = fmaf(xv, y3.x, acc[12]); acc[13] = fmaf(xv, y3.y, acc[13]); acc[14] = fmaf(xv, y3.z, acc[14]); acc[15] = fmaf(xv, y3.w, acc[15]);
        }
    }
    if (ci < d.ci) {
        float* o = a.slab + (((long)split * d.w_taps + d.tap_w[tap]) * 64 + ci) * 64 + co8;
#pragma unroll
        for (int e = 0; e < 16; e += 4)
            if (co8 + e < d.co) *reinterpret_cast<float4*>(o + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
    }
}

// bias_slab[row][64] = column sums of dy over the pixels of chunk `row` (fixed order inside a chunk, rows added by the reduction)
__global__ __launch_bounds__(256) void dy_colsum_kernel(const float* __restrict__ dy, long n_pix, int pix_stride, int co, long per_block,
                                                        float* __restrict__ bias_slab) {
    __shared__ float4 red4[256];
    const long p0 = (long)blockIdx.x * per_block, p1 = p0 + per_block < n_pix ? p0 + per_block : n_pix;
    if ((co & 3) == 0 && (pix_stride & 3) == 0 && (((uintptr_t)dy) & 15) == 0) {
        // 16-byte loads: co / 4 lanes per pixel, 256 / (co / 4) pixels per round, four independent loads in flight (the former loop -- one
        // 4-byte load per lane and round, half the wave idle at 32 channels -- took 32 us for 16.8 MB)
        const int lanes = co >> 2, rowsl = 256 / lanes;
        const int c4 = threadIdx.x % lanes, pr = threadIdx.x / lanes;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pr < rowsl) {
            long p = p0 + pr;
            for (; p + 3L * rowsl < p1; p += 4L * rowsl) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(dy + (p + (long)u * rowsl) * pix_stride + c4 * 4);
#pragma unroll
                for (int u = 0; u < 4; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            for (; p < p1; p += rowsl) {
                const float4 v = *reinterpret_cast<const float4*>(dy + p * pix_stride + c4 * 4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        red4[threadIdx.x] = acc;
        __syncthreads();
        if ((int)threadIdx.x < 64) {
            float s_ = 0.0f;
            if ((int)threadIdx.x < co) {
                const int l = threadIdx.x >> 2, e = threadIdx.x & 3;
                for (int q = 0; q < rowsl; q++) {  // the pixel lanes' sums in lane order
                    const float4 v = red4[q * lanes + l];
                    s_ += e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
                }
            }
            bias_slab[(long)blockIdx.x * 64 + threadIdx.x] = s_;
        }
        return;
    }
    float* red = reinterpret_cast<float*>(red4);  // [4][64]
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    float s = 0.0f;
    if (c < co)
        for (long p = p0 + part; p < p1; p += 4) s += dy[p * pix_stride + c];
    red[part * 64 + c] = s;
    __syncthreads();
    if (part == 0) bias_slab[(long)blockIdx.x * 64 + c] = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
}

constexpr int kSpBiasRows = 256;

size_t sparse_layout(const liso_conv_desc& d, size_t* off_rowcnt, size_t* off_rowoff, size_t* off_cells, size_t* off_slab, size_t* off_bias) {
    const size_t rows = (size_t)d.batch * d.hi;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t at = o;
        o += (bytes + 255) / 256 * 256;
        return at;
    };
    *off_rowcnt = take(rows * 4);
    *off_rowoff = take((rows + 1) * 4);
    *off_cells = take(rows * d.wi * 4);  // (worst case: every cell occupied)
    *off_slab = take((size_t)kSpSplits * d.w_taps * 64 * 64 * 4);
    *off_bias = take((size_t)kSpBiasRows * 64 * 4);
    return o;
}

bool sparse_ok(const liso_conv_desc& d) {
    return d.batch > 0 && d.n_classes == 1 && d.osy == 1 && d.osx == 1 && d.ci <= 64 && d.co <= 64 && d.ci % 4 == 0 && d.co % 4 == 0 &&
           d.x_pix_stride % 4 == 0 && d.n_taps >= 1 && d.n_taps <= LISO_CONV_MAX_TAPS && d.n_taps == d.w_taps &&
           (d.mode == LISO_CONV_F32X3 || d.mode == LISO_CONV_F32) && (long)d.batch * d.hi * d.wi < (1l << 31) &&
           (long)d.batch * d.hi <= 1024l * 64 && d.in_affine_batch_stride == 0;
}

// ---- plan of the row-stationary 3x3 kernel ----------------------------------------------------------------------------------------
struct Rs3Plan {
    int th, stride, lds, blocks;
    size_t slab_bytes, bias_bytes;
    WgArgs a;
};

bool make_rs3_plan(const liso_conv_desc& d, Rs3Plan* p) {
    if (const char* e = getenv("LISO_WGRAD_RS3"))  // experiments / A-B runs: 0 = the generic kernel everywhere
        if (atoi(e) == 0) return false;
    if ((d.mode != LISO_CONV_BF16 && d.mode != LISO_CONV_F32X3) || d.n_classes != 1 || d.n_taps != 9 || d.w_taps != 9) return false;
    const bool x3 = d.mode == LISO_CONV_F32X3;
    const int vec = x3 ? 4 : 8;
    const int S = d.isy;  // 3x3, padding 1, stride 1 or 2 (stride 2: bf16 only -- two buffers of two planes would not fit)
    if ((S != 1 && S != 2) || d.isx != S || d.osy != 1 || d.osx != 1 || d.in_affine_batch_stride != 0) return false;
    if (S == 2 && (x3 || getenv("LISO_WGRAD_RS3_S2_OFF"))) return false;
    if (d.hv != d.ho || d.wv != d.wo || d.ho != (d.hi - 1) / S + 1 || d.wo != (d.wi - 1) / S + 1 || d.batch <= 0) return false;
    if (d.ci % vec || d.co % vec || d.x_pix_stride % vec || d.class_tap_begin[0] != 0 || d.class_tap_begin[1] != 9) return false;
    for (int t = 0; t < 9; t++)
        if (d.tap_dy[t] != t / 3 - 1 || d.tap_dx[t] != t % 3 - 1 || d.tap_w[t] < 0 || d.tap_w[t] >= 9) return false;
    WgArgs& a = p->a;
    // 32 input channels per block (half the slab per launch and block, twice the tiles per block) -- only where the layer HAS no more than
    // 32 (the SLIM encoders' first stage: 42.4 -> 38.5 us).  Measured (scripts/wgrad_time.py, wgrad + reduction): bf16 4 x 128 -> 128 at
    // 128^2 39.3 -> 49.1 us, B = 2 32.3 -> 45.9, 64 -> 64 at 256^2 39.4 -> 48.5, 256 -> 256 at 64^2 40.8 -> 51.1; fp32 64 -> 64 35.7 ->
    // 37.3, 304 -> 192 x 12 maps 192.7 -> 227.8: a wave that multiplies one column half per tile row pays the tile's barrier and the
    // priming of its row-stationary fragments for half the products -- the slab bytes it saves are worth less.  LISO_WGRAD_CIW: experiments
    a.ci_w = (S == 1 && d.ci <= 32) ? 32 : CT;
    if (const char* e = getenv("LISO_WGRAD_CIW")) a.ci_w = (atoi(e) == 32 && S == 1) ? 32 : CT;
    a.ci_t = (d.ci + a.ci_w - 1) / a.ci_w;
    a.co_t = (d.co + CT - 1) / CT;
    const long cc = (long)a.ci_t * a.co_t;
    long want = 256 / cc;  // one block per CU
    if (const char* e = getenv("LISO_WGRAD_BLOCKS")) want = (atol(e) > 0 ? atol(e) : 256) / cc;  // experiments
    want = want < 1 ? 1 : want;
    a.tiles_x = (d.wo + TW - 1) / TW;
    const long tiles8 = (long)d.batch * ((d.ho + 7) / 8) * a.tiles_x;
    int th = tiles8 >= 4 * want ? 8 : 4;
    if (const char* e = getenv("LISO_WGRAD_TH")) th = atoi(e) == 8 ? 8 : atoi(e) == 4 ? 4 : th;  // experiments
    if (x3 || S == 2) th = 3;  // (two planes per operand / the (2 TH + 1) x 65 halo of stride 2: two buffers of a 3-row tile fit the LDS)
    p->stride = S;
    a.th = th;
    a.tiles_y = (d.ho + th - 1) / th;
    a.n_tiles = d.batch * a.tiles_y * a.tiles_x;
    long min_tiles = 4;  // >= 4 tiles per block: the first tile's load is exposed, the others hide behind MFMAs
    // (experiments; 2 tiles per block on the SLIM encoders' 64 -> 64 / 96 -> 96 layers -- 172 blocks instead of 86 -- measured no change:
    // SLIM step 10.81-11.06 vs 10.66-11.05 ms, twice the slabs)
    if (const char* e = getenv("LISO_WGRAD_RS3_MIN_TILES")) min_tiles = atol(e) > 0 ? atol(e) : min_tiles;
    long s = a.n_tiles / min_tiles;
    s = s > want ? want : s;
    s = s < 1 ? 1 : s;
    a.splits = (int)s;
    a.n_groups = 1;
    p->th = th;
    p->blocks = (int)(cc * s);
    const int npx = ((th - 1) * S + 3) * ((TW - 1) * S + 3);
    p->lds = 2 * (x3 ? 2 : 1) * ((npx + 3) / 4 * 4 + th * TW) * RPS;  // two tile buffers (of two planes each for F32X3)
    p->slab_bytes = (size_t)s * 9 * a.ci_t * a.ci_w * a.co_t * CT * sizeof(float);
    p->bias_bytes = (size_t)s * a.co_t * CT * sizeof(float);
    return true;
}

#ifdef LISO_WGRAD_STAMPS
unsigned long long* g_wgrad_stamps = nullptr;  // 4096 blocks x 16 counters
#endif

template <int MODE, int TH, int S, int CIW = 64>
int launch_rs3(const liso_conv_desc& d, const Rs3Plan& p, hipStream_t st) {
    static liso_dev::PerDeviceFlag attr_set;
    if (!liso_dev::lds_opt_in(attr_set, (const void*)conv_wgrad_rs3_kernel<MODE, TH, S, CIW>, 160 * 1024)) return LISO_ELAUNCH;
#ifdef LISO_WGRAD_STAMPS
    if (!g_wgrad_stamps && hipMalloc((void**)&g_wgrad_stamps, 4096 * 16 * 8) != hipSuccess) return LISO_ELAUNCH;
    (void)hipMemsetAsync(g_wgrad_stamps, 0, 4096 * 16 * 8, st);
    Rs3Plan q = p;
    q.a.stamps = p.blocks <= 4096 ? g_wgrad_stamps : nullptr;
    conv_wgrad_rs3_kernel<MODE, TH, S, CIW><<<q.blocks, kRsThreads, q.lds, st>>>(d, q.a);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
#endif
    conv_wgrad_rs3_kernel<MODE, TH, S, CIW><<<p.blocks, kRsThreads, p.lds, st>>>(d, p.a);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // namespace

// ---- weight gradient of a k x k / stride-1 convolution with at most FOUR input channels (fp32) -----------------------------------
// The motion encoder's 7x7 layers read the 2-channel flow / 4-channel class logits (liso/slim/model/update.py:57,66).  As an MFMA
// problem that is a [196 x pixels] . [pixels x 64] product whose 196 rows are shifted copies of 4 numbers per pixel: padding it to
// the 64-channel tiles of the kernels above costs 0.33 ms.  Here a wave owns ONE input channel and its 64 lanes are 64 output
// channels; the lane keeps the k x k accumulators of its (ci, co) pair and a k x k window of the input around the current pixel in
// registers, and walks a row: per pixel k new window values (wave-uniform LDS reads), one dy value (coalesced 256-B row) and k*k
// exact fp32 FMAs -- no tile padding, no im2col.  Blocks own whole rows; their partial filters are summed in block order by a second
// launch (bitwise reproducible).  0.6 GFLOP for the deferred batch of 12 maps at 64^2.
constexpr int kSmallCi = 4;

template <int K>
__global__ __launch_bounds__(256) void wgrad_smallci_kernel(const float* __restrict__ x, long xps, const float* __restrict__ dy, long gps,
                                                            int batch, int h, int w, int co, int rows_per_block,
                                                            float* __restrict__ partial, float* __restrict__ bias_partial) {
    extern __shared__ float4 xs[];  // [K][w + K - 1] pixels x 4 channels, zero outside the map; then the dy row [w][64]
    constexpr int P = K / 2;
    const int lane = threadIdx.x & 63, ci = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;  // output channel of this lane
    const int wp = w + K - 1;
    float* gs = reinterpret_cast<float*>(xs + K * wp);
    float acc[K][K];
#pragma unroll
    for (int a = 0; a < K; a++)
#pragma unroll
        for (int b = 0; b < K; b++) acc[a][b] = 0.f;
    float bsum = 0.f;
    const long rows = (long)batch * h;
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const float* xs_f = reinterpret_cast<const float*>(xs);
    for (long r = r0; r < r1; r++) {
        const int b = (int)(r / h), y = (int)(r % h);
        __syncthreads();
        // (both stagings are independent loads: one round trip per row, not one per window step)
        const float* g = dy + (((long)b * h + y) * w) * gps;
        for (int i = threadIdx.x; i < w * 64; i += 256) {
            const int px = i >> 6, cc = blockIdx.y * 64 + (i & 63);
            gs[i] = cc < co ? g[(long)px * gps + cc] : 0.f;
        }
        for (int i = threadIdx.x; i < K * wp; i += 256) {
            const int ky = i / wp, px = i % wp;
            const int yy = y + ky - P, xx = px - P;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = *reinterpret_cast<const float4*>(x + (((long)b * h + yy) * w + xx) * xps);
            xs[i] = v;
        }
        __syncthreads();
        float win[K][K];  // win[ky][slot]: padded column col lives in slot col % K
#pragma unroll
        for (int ky = 0; ky < K; ky++)
#pragma unroll
            for (int col = 0; col < K - 1; col++) win[ky][col] = xs_f[(ky * wp + col) * 4 + ci];
        for (int px0 = 0; px0 < w; px0 += K) {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const int col = px0 + j + K - 1;  // the column that enters the window of pixel px0 + j; px0 is a multiple of K
                const int cc = col < wp ? col : wp - 1;  // (beyond the row: its products meet a zero below)
                const float gv = px0 + j < w ? gs[(px0 + j) * 64 + lane] : 0.f;
#pragma unroll
                for (int ky = 0; ky < K; ky++) win[ky][(j + K - 1) % K] = xs_f[(ky * wp + cc) * 4 + ci];
#pragma unroll
                for (int ky = 0; ky < K; ky++)
#pragma unroll
                    for (int kx = 0; kx < K; kx++) acc[ky][kx] = fmaf(win[ky][(j + kx) % K], gv, acc[ky][kx]);
                bsum += gv;
            }
        }
    }
    const int cop = gridDim.y * 64;
    float* pt = partial + (((size_t)blockIdx.x * kSmallCi + ci) * K * K) * cop + c;
#pragma unroll
    for (int ky = 0; ky < K; ky++)
#pragma unroll
        for (int kx = 0; kx < K; kx++) pt[(size_t)(ky * K + kx) * cop] = acc[ky][kx];
    if (ci == 0 && bias_partial) bias_partial[(size_t)blockIdx.x * cop + c] = bsum;
}

// dw[co][ci][k][k] = sum over blocks of partial[block][ci][tap][co]; the bias likewise.  64 outputs x 8 block groups per workgroup:
// group q adds the blocks q, q + 8, ... (8 independent loads in flight per thread), the groups' sums are added in group order
// through LDS -- a fixed order, and 8 x 8 loads deep instead of one dependent chain over all blocks (measured: 384 blocks in one
// chain per output took longer than the gradient kernel itself).
__global__ __launch_bounds__(512) void wgrad_smallci_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias_partial,
                                                                   int nblk, int kk, int co, int cop, float* __restrict__ dw,
                                                                   float* __restrict__ db) {
    __shared__ float sm[8][64];
    const int i = blockIdx.x * 64 + threadIdx.x, q = threadIdx.y;
    const int n = kSmallCi * kk * cop;
    const float* src = nullptr;
    size_t stride = 0;
    if (i < n) { src = partial + i; stride = (size_t)n; }
    else if (bias_partial && i - n < cop) { src = bias_partial + (i - n); stride = (size_t)cop; }
    float s = 0.f;
    if (src) {
        int b = q;
        for (; b + 56 < nblk; b += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)(b + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; u++) s += v[u];
        }
        for (; b < nblk; b += 8) s += src[(size_t)b * stride];
    }
    sm[q][threadIdx.x] = s;
    __syncthreads();
    if (q != 0 || !src) return;
    s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; u++) s += sm[u][threadIdx.x];
    if (i < n) {
        const int c = i % cop, t = (i / cop) % kk, ci = i / (cop * kk);
        if (c < co) dw[((size_t)c * kSmallCi + ci) * kk + t] = s;
    } else if (db && i - n < co) {
        db[i - n] = s;
    }
}

inline bool smallci_layout(int batch, int h, int w, int co, int k, int* nblk, int* rpb, size_t* part_bytes, size_t* bias_bytes, int* lds) {
    if (batch < 1 || h < 1 || w < 1 || co < 1 || co > 1024 || (k != 3 && k != 5 && k != 7)) return false;
    const long rows = (long)batch * h;
    long nb = rows < 512 ? rows : 512;  // (24 KB of LDS and 158 VGPRs: several blocks per CU hide each other's staging)
    const long per = (rows + nb - 1) / nb;
    nb = (rows + per - 1) / per;
    const int cop = (co + 63) / 64 * 64;
    *nblk = (int)nb;
    *rpb = (int)per;
    *part_bytes = (size_t)nb * kSmallCi * k * k * cop * sizeof(float);
    *bias_bytes = (size_t)nb * cop * sizeof(float);
    *lds = k * (w + k - 1) * (int)sizeof(float4) + w * 64 * (int)sizeof(float);
    return *lds <= 64 * 1024;
}


extern "C" {

size_t liso_conv_wgrad_workspace_bytes(const liso_conv_desc* d) {
    if (!d) return 0;
    Rs3Plan r;
    if (make_rs3_plan(*d, &r)) return r.slab_bytes + r.bias_bytes;
    WgPlan p;
    if (!make_plan(*d, &p)) return 0;
    return p.slab_bytes + p.bias_bytes;
}

size_t liso_conv_wgrad_sparse_workspace_bytes(const liso_conv_desc* d) {
    if (!d || !sparse_ok(*d)) return 0;
    size_t a, b, c, e, f;
    return sparse_layout(*d, &a, &b, &c, &e, &f);
}

int liso_conv_wgrad_sparse_f32(const liso_conv_desc* d, const float* x, const float* occupancy, const float* dy, int dy_pix_stride,
                               float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !x || !occupancy || !dy || !dw || !workspace) return LISO_EINVAL;
    if (!sparse_ok(*d) || dy_pix_stride % 4 || dy_pix_stride < d->co || (((uintptr_t)x | (uintptr_t)dy) & 15)) return LISO_EINVAL;
    size_t o_cnt, o_off, o_cells, o_slab, o_bias;
    if (workspace_bytes < sparse_layout(*d, &o_cnt, &o_off, &o_cells, &o_slab, &o_bias)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int rows = d->batch * d->hi;
    cells_count_kernel<<<rows, 256, 0, st>>>(occupancy, d->wi, (int*)(ws + o_cnt));
    cells_scan_kernel<<<1, 1024, 0, st>>>((const int*)(ws + o_cnt), rows, (int*)(ws + o_off));
    cells_fill_kernel<<<rows, 64, 0, st>>>(occupancy, d->wi, (const int*)(ws + o_off), (int*)(ws + o_cells));
    SpArgs a;
    a.x = x;
    a.dy = dy;
    a.cells = (const int*)(ws + o_cells);
    a.n_cells = (const int*)(ws + o_off) + rows;
    a.slab = (float*)(ws + o_slab);
    a.dy_pix_stride = dy_pix_stride;
    a.pad = 0;
    wgrad_sparse_kernel<<<dim3(d->n_taps, kSpSplits), 256, 0, st>>>(*d, a);
    if (dbias) {
        const long n_pix = (long)d->batch * d->ho * d->wo;
        const long per = (n_pix + kSpBiasRows - 1) / kSpBiasRows;
        dy_colsum_kernel<<<kSpBiasRows, 256, 0, st>>>(dy, n_pix, dy_pix_stride, d->co, per, (float*)(ws + o_bias));
    }
    if (hipGetLastError() != hipSuccess) return LISO_ELAUNCH;
    return launch_reduce((const float*)(ws + o_slab), (const float*)(ws + o_bias), kSpSplits, kSpBiasRows, d->w_taps, d->ci, d->co, 64, 64, 0,
                         dw, dbias, st);
}

int liso_conv_wgrad(const liso_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, const void* dy,
                    int dy_pix_stride, int transposed, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!d || !x || !dy || !dw || !workspace) return LISO_EINVAL;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return LISO_EINVAL;
    const int vec = d->mode == LISO_CONV_BF16 ? 8 : 4;
    if (dy_pix_stride % vec || dy_pix_stride < d->co || (((uintptr_t)x | (uintptr_t)dy) & 15)) return LISO_EINVAL;
    if (d->wgrad_co < 0 || d->wgrad_co > d->co) return LISO_EINVAL;
    Rs3Plan r3;
    if (make_rs3_plan(*d, &r3)) {  // 3x3 / stride 1, bf16: all taps per block, row-stationary fragments
        if (workspace_bytes < r3.slab_bytes + r3.bias_bytes) return LISO_EWORKSPACE;
        r3.a.x = x;
        r3.a.in_scale = in_scale;
        r3.a.in_shift = in_shift;
        r3.a.dy = dy;
        r3.a.dy_pix_stride = dy_pix_stride;
        r3.a.slab = (float*)workspace;
        r3.a.bias_slab = dbias ? (float*)((char*)workspace + r3.slab_bytes) : nullptr;
        hipStream_t st3 = (hipStream_t)stream;
        int rc3;
        if (r3.a.ci_w == 32)
            rc3 = d->mode == LISO_CONV_F32X3 ? launch_rs3<LISO_CONV_F32X3, 3, 1, 32>(*d, r3, st3)
                  : r3.th == 8           ? launch_rs3<LISO_CONV_BF16, 8, 1, 32>(*d, r3, st3)
                                         : launch_rs3<LISO_CONV_BF16, 4, 1, 32>(*d, r3, st3);
        else
            rc3 = d->mode == LISO_CONV_F32X3 ? launch_rs3<LISO_CONV_F32X3, 3, 1>(*d, r3, st3)
                  : r3.stride == 2       ? launch_rs3<LISO_CONV_BF16, 3, 2>(*d, r3, st3)
                  : r3.th == 8           ? launch_rs3<LISO_CONV_BF16, 8, 1>(*d, r3, st3)
                                         : launch_rs3<LISO_CONV_BF16, 4, 1>(*d, r3, st3);
        if (rc3 != LISO_OK) return rc3;
        const int co_w3 = d->wgrad_co > 0 ? d->wgrad_co : d->co;
        return launch_reduce(r3.a.slab, r3.a.bias_slab, r3.a.splits, r3.a.splits, d->w_taps, d->ci, co_w3, (long)r3.a.ci_t * r3.a.ci_w,
                             (long)r3.a.co_t * CT, transposed, dw, dbias, st3);
    }
    WgPlan p;
    if (!make_plan(*d, &p)) return LISO_EINVAL;
    if (workspace_bytes < p.slab_bytes + p.bias_bytes) return LISO_EWORKSPACE;
    p.a.x = x;
    p.a.in_scale = in_scale;
    p.a.in_shift = in_shift;
    p.a.dy = dy;
    p.a.dy_pix_stride = dy_pix_stride;
    p.a.slab = (float*)workspace;
    p.a.bias_slab = dbias ? (float*)((char*)workspace + p.slab_bytes) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (d->mode == LISO_CONV_F32)
        rc = p.tg == 7 ? launch<LISO_CONV_F32, 7>(*d, p, st) : p.tg == 3 ? launch<LISO_CONV_F32, 3>(*d, p, st) : launch<LISO_CONV_F32, 1>(*d, p, st);
    else if (d->mode == LISO_CONV_F32X3)
        rc = p.tg == 7   ? launch<LISO_CONV_F32X3, 7>(*d, p, st)
             : p.tg == 3 ? launch<LISO_CONV_F32X3, 3>(*d, p, st)
                         : launch<LISO_CONV_F32X3, 1>(*d, p, st);
    else
        rc = p.tg == 9   ? launch<LISO_CONV_BF16, 9>(*d, p, st)
             : p.tg == 7 ? launch<LISO_CONV_BF16, 7>(*d, p, st)
             : p.tg == 3 ? launch<LISO_CONV_BF16, 3>(*d, p, st)
                         : launch<LISO_CONV_BF16, 1>(*d, p, st);
    if (rc != LISO_OK) return rc;
    if (d->wgrad_co < 0 || d->wgrad_co > d->co) return LISO_EINVAL;
    const int co_w = d->wgrad_co > 0 ? d->wgrad_co : d->co;  // channels written (dy may carry zero-padded channels beyond)
    return launch_reduce(p.a.slab, p.a.bias_slab, p.splits, p.splits * d->n_classes, d->w_taps, d->ci, co_w, (long)p.a.ci_t * CT,
                         (long)p.a.co_t * CT, transposed, dw, dbias, st);
}

size_t liso_conv_wgrad_smallci_workspace_bytes(int batch, int h, int w, int co, int k) {
    int nblk, rpb, lds;
    size_t pb, bb;
    if (!smallci_layout(batch, h, w, co, k, &nblk, &rpb, &pb, &bb, &lds)) return 0;
    return pb + bb;
}

int liso_conv_wgrad_smallci_f32(const float* x, long x_pix_stride, const float* dy, long dy_pix_stride, int batch, int h, int w, int co,
                                int k, float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    int nblk, rpb, lds;
    size_t pb, bb;
    if (!x || !dy || !dw || !workspace || x_pix_stride < kSmallCi || dy_pix_stride < co || (((uintptr_t)x) & 15) || (x_pix_stride & 3))
        return LISO_EINVAL;
    if (!smallci_layout(batch, h, w, co, k, &nblk, &rpb, &pb, &bb, &lds)) return LISO_EINVAL;
    if (workspace_bytes < pb + bb) return LISO_EWORKSPACE;
    float* partial = (float*)workspace;
    float* bias_partial = dbias ? (float*)((char*)workspace + pb) : nullptr;
    const int tiles = (co + 63) / 64, cop = tiles * 64;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nblk, (unsigned)tiles);
#define LISO_SMALLCI(K) \
    hipLaunchKernelGGL(wgrad_smallci_kernel<K>, grid, dim3(256), lds, st, x, x_pix_stride, dy, dy_pix_stride, batch, h, w, co, rpb, partial, bias_partial)
    if (k == 7) LISO_SMALLCI(7);
    else if (k == 5) LISO_SMALLCI(5);
    else LISO_SMALLCI(3);
#undef LISO_SMALLCI
    const int n = kSmallCi * k * k * cop + (dbias ? cop : 0);
    hipLaunchKernelGGL(wgrad_smallci_reduce_kernel, dim3((n + 63) / 64), dim3(64, 8), 0, st, partial, bias_partial, nblk, k * k, co, cop, dw, dbias);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // extern "C"

#ifdef LISO_WGRAD_STAMPS
// diagnostic build only: the counters of the last conv_wgrad_rs3_kernel launch -> host (16 x uint64 per block)
extern "C" int liso_wgrad_stamps_read(unsigned long long* host_out, int blocks) {
    if (!g_wgrad_stamps || blocks > 4096) return LISO_EINVAL;
    return hipMemcpy(host_out, g_wgrad_stamps, (size_t)blocks * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}
#endif
