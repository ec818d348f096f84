// On-the-fly RAFT correlation lookup for gfx950 (MI355X).  C ABI + reference lines: include/liso_slim.h.
//
// One wavefront owns one (query pixel, pyramid level).  Because the 7x7 window sits at integer offsets around ONE
// fractional centre, all 49 bilinear samples are combinations of the dot products with an 8x8 integer patch:
//     P[u][v] = < f1 , f2_i[y0+v][x0+u] >,   x0 = floor(cx/2^i) - r,  y0 = floor(cy/2^i) - r      (0 outside the map)
//     out[a][b] = (1-fx)(1-fy) P[a][b] + fx(1-fy) P[a+1][b] + (1-fx)fy P[a][b+1] + fx fy P[a+1][b+1]
// Lane mapping: 32 lanes x float4 cover the D=128 channels of one patch pixel (one coalesced 512-B row read per
// half-wave), the two half-waves stream two patch pixels at once; the 32 partial sums per lane are reduced with a
// transposed butterfly (31 cross-lane moves for 32 values instead of 160), leaving lane l with P[2*(l&31)+(l>>5)].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"
#include "per_device.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kTransposeStride = 36;  // floats per lane row in the LDS transpose (32 + 4: 16-B aligned, conflict-free columns)

struct LevelPtrs {
    const float* f2[LISO_CORR_MAX_LEVELS];
    float* g2[LISO_CORR_MAX_LEVELS];
};

struct Patch {
    int x0, y0, H, W;
    float fx, fy;
};

__device__ __forceinline__ Patch patch_of(const liso_corr_cfg& c, const float* __restrict__ coords, int b, int pix, int lvl) {
    Patch p;
    const int hw = c.h * c.w;
    // corr.py:37: centroid_lvl = coords / 2**i ; grid_sample(align_corners=True) maps it back to pixel units
    const float scale = 1.0f / (float)(1 << lvl);
    const float cx = coords[((size_t)b * 2 + 0) * hw + pix] * scale;
    const float cy = coords[((size_t)b * 2 + 1) * hw + pix] * scale;
    const float flx = floorf(cx), fly = floorf(cy);
    p.fx = cx - flx; p.fy = cy - fly;
    p.x0 = (int)flx - c.radius; p.y0 = (int)fly - c.radius;
    p.H = c.h >> lvl; p.W = c.w >> lvl;
    return p;
}

// one (query pixel, level) by one wavefront; Pw: 64 floats, Tw: 64 x kTransposeStride floats of LDS owned by the wave
template <int VEC>  // VEC float4 per lane: D = 128 * VEC
__device__ __forceinline__ void lookup_one(const liso_corr_cfg& c, const float* __restrict__ fmap1, const LevelPtrs& lp,
                                           const float* __restrict__ coords, float* __restrict__ out, int b, int pix, int lvl, int lane,
                                           float* Pw, float (*Tw)[kTransposeStride]) {
    const int hw = c.h * c.w;
    const Patch pt = patch_of(c, coords, b, pix, lvl);
    const int g = lane & 31, half = lane >> 5;
    const int D = c.dim;
    float4 f1[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) f1[k] = *reinterpret_cast<const float4*>(fmap1 + ((size_t)b * hw + pix) * D + (k * 32 + g) * 4);
    const float* f2 = lp.f2[lvl] + (size_t)b * pt.H * pt.W * D;
    float acc[32];
    // Loads are unconditional (coordinates clamped into the map, the result masked afterwards) and issued 8 patch pixels
    // at a time: a branch around each load would serialise 32 dependent L2 round trips per wavefront.
    constexpr int kBatch = 8;
#pragma unroll
    for (int t0 = 0; t0 < 32; t0 += kBatch) {
        float4 v[kBatch][VEC];
        bool inside[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            const int q = 2 * (t0 + j) + half;  // patch index: u = q & 7 (x), v = q >> 3 (y)
            const int x = pt.x0 + (q & 7), y = pt.y0 + (q >> 3);
            inside[j] = x >= 0 && x < pt.W && y >= 0 && y < pt.H;  // zeros padding of grid_sample
            const int xc = min(max(x, 0), pt.W - 1), yc = min(max(y, 0), pt.H - 1);
            const float* row = f2 + ((size_t)yc * pt.W + xc) * D;
#pragma unroll
            for (int k = 0; k < VEC; k++) v[j][k] = *reinterpret_cast<const float4*>(row + (k * 32 + g) * 4);
        }
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; k++)
                s += f1[k].x * v[j][k].x + f1[k].y * v[j][k].y + f1[k].z * v[j][k].z + f1[k].w * v[j][k].w;
            acc[t0 + j] = inside[j] ? s : 0.f;
        }
    }
    // Sum over the 32 lanes of each half-wave of 32 per-lane values, lane g keeping the total of acc[g]: a transpose through
    // LDS (8 x 16-B writes per lane, then 32 conflict-free 4-B reads down a column; rows padded to 36 floats).  The
    // 31-shuffle butterfly this replaces (ds_bpermute) took 97 of the kernel's 123 us: measured 26 us without it.
    {
        float4* wrow = reinterpret_cast<float4*>(&Tw[lane][0]);
#pragma unroll
        for (int k = 0; k < 8; k++) wrow[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
    }
    __builtin_amdgcn_wave_barrier();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 32; j++) tot += Tw[half * 32 + j][g];  // fixed order: bit reproducible
    Pw[2 * g + half] = tot;
    __builtin_amdgcn_wave_barrier();
    const int W7 = 2 * c.radius + 1;
    const int a = lane / W7, bb = lane % W7;  // a: x offset index, bb: y offset index (corr.py:31-35 order)
    if (lane < W7 * W7) {
        const float inv = 1.0f / sqrtf((float)D);  // corr.py:56
        const float p00 = Pw[bb * 8 + a], p10 = Pw[bb * 8 + a + 1];
        const float p01 = Pw[(bb + 1) * 8 + a], p11 = Pw[(bb + 1) * 8 + a + 1];
        const float v = (1.f - pt.fx) * (1.f - pt.fy) * p00 + pt.fx * (1.f - pt.fy) * p10 + (1.f - pt.fx) * pt.fy * p01 +
                        pt.fx * pt.fy * p11;
        const int C = c.levels * W7 * W7;
        out[((size_t)b * hw + pix) * C + lvl * W7 * W7 + lane] = v * inv;
    }
}

template <int VEC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void corr_lookup_fwd_kernel(liso_corr_cfg c, const float* __restrict__ fmap1, LevelPtrs lp,
                                                                              const float* __restrict__ coords, float* __restrict__ out) {
    __shared__ float P[kWavesPerBlock][64];
    __shared__ float T[kWavesPerBlock][64][kTransposeStride];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hw = c.h * c.w;
    // XCD-aware block order: the hardware deals consecutive workgroups round-robin to the 8 XCDs, each with its own 4 MB
    // L2.  Logical block = (physical % 8) * (blocks / 8) + physical / 8 gives every XCD one contiguous eighth of the
    // (sample, pixel) range: its level-0 working set is then ~1/4 of one sample's feature map (0.5 MB + halo) instead of
    // all 5.3 MB of both samples' pyramids, i.e. L2 resident.  (gridDim.x is a multiple of 8.)
    const long per_xcd = gridDim.x / 8;
    const long block = (long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const long item = block * kWavesPerBlock + wave;  // (b, pix, lvl)
    const long total = (long)c.batch * hw * c.levels;
    if (item >= total) return;
    const int lvl = (int)(item % c.levels);
    const int pix = (int)((item / c.levels) % hw);
    const int b = (int)(item / ((long)c.levels * hw));
    lookup_one<VEC>(c, fmap1, lp, coords, out, b, pix, lvl, lane, P[wave], T[wave]);
}

// ---- tiled form: 4 x 8 neighbouring query pixels of one level share the feature rows they correlate with ---------------------------------
// The per-query kernel above reads an 8 x 8 patch of D-channel rows per (query, level): 32 KB (D = 128) through L1 / L2 for 16 K flops,
// 2.1 GB per lookup of 4 x 64 x 64 queries x 4 levels (84 us: the largest single kernel of the SLIM inference replay).  Neighbouring queries
// have neighbouring centres -- the flow field is smooth at 1/8 resolution --, so the patches of a 4 x 8 block of queries cover a region of
// (4 + 7 + spread) x (8 + 7 + spread) rows, 165 at level 0 and fewer on the pooled levels: this kernel stages that region ONCE per block,
//     C[q][p] = < f1[q] , f2[p] >   for the block's 32 queries x the region's <= 256 rows
// on the matrix cores (fp32 operands as bf16 hi / lo pairs, hi hi + hi lo + lo hi: the arithmetic of the F32X3 convolutions around it), and
// every query then reads its own 8 x 8 window out of C in LDS for the 49 bilinear samples (same formula as above).  Rows outside the
// map are zero (grid_sample's padding); the region is clipped to the map.  A block whose region would exceed 256 rows (queries whose
// centres lie far apart: garbage flow, never seen on real sweeps) takes the per-query path, one wave per query.
// LDS: A planes 2 x 4 KB ([k8][query][8 bf16]) | B planes 2 x 32 KB ([k8][row][8 bf16]) per 64-channel chunk; C (32 x 260 fp32) and the
// fallback's transpose buffers alias the B planes.
// MT = 32-query tiles per block: 4 x 8 queries on level 0 (MT 1, regions of <= 256 rows), 8 x 8 on the pooled levels (MT 2, <= 192 rows: at
// half / quarter / eighth resolution 64 queries cover hardly more rows than 32, and a block per 64 queries halves the blocks, the
// staged bytes per query and the exposed load phases of those levels).
constexpr int kTQW = 8, kTKC = 64;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 cbf8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float cf16v;

// (native conversions: v_cvt_pk_bf16_f32, round to nearest even -- the convolutions' split; bit arithmetic here made the split the
// kernel's bottleneck: ~70 vector instructions per 16 bytes against ~14)
__device__ __forceinline__ unsigned corr_pack_bf16(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
// 4 floats -> 8 B of the hi plane + 8 B of the lo plane
__device__ __forceinline__ void corr_split4(const float4 v, uint2* hi, uint2* lo) {
    const unsigned p0 = corr_pack_bf16(v.x, v.y), p1 = corr_pack_bf16(v.z, v.w);
    *hi = make_uint2(p0, p1);
    *lo = make_uint2(corr_pack_bf16(v.x - __uint_as_float(p0 << 16), v.y - __uint_as_float(p0 & 0xffff0000u)),
                     corr_pack_bf16(v.z - __uint_as_float(p1 << 16), v.w - __uint_as_float(p1 & 0xffff0000u)));
}

template <int VEC, int MT, int RMAX>
__device__ __forceinline__ void corr_tile_body(const liso_corr_cfg& c, const float* __restrict__ fmap1, const LevelPtrs& lp,
                                               const float* __restrict__ coords, float* __restrict__ out, int tiles_x, int tiles_y, int lvl0,
                                               int n_lvl, long item, unsigned char* lds) {
    constexpr int D = 128 * VEC;
    constexpr int kTQH = 4 * MT, kTQ = 32 * MT, kTRMax = RMAX, kTCStride = RMAX + 4;
    // bytes of one 8-channel group of a plane: rows x 16 B + 64 B, so that the eight groups a wave's split stores touch at once start 64 B
    // apart in the banks (unpadded they all start on the same bank: PMC round 5, 0.69 of the LDS cycles were conflict cycles)
    constexpr int AS = kTQ * 16 + 64, BS = kTRMax * 16 + 64;
    constexpr int kTLdsA = 2 * (kTKC / 8) * AS, kTLdsB = 2 * (kTKC / 8) * BS;
    static_assert(kTQ * kTCStride * 4 <= kTLdsA + kTLdsB, "C does not fit the operand buffers");
    __shared__ int q_x0[kTQ], q_y0[kTQ], q_pix[kTQ];
    __shared__ float q_fx[kTQ], q_fy[kTQ];
    __shared__ int reg[4];  // region: x0, y0, width, height
    __shared__ float Pf[kWavesPerBlock][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = c.h * c.w;
    const int tx = (int)(item % tiles_x);
    item /= tiles_x;
    const int ty = (int)(item % tiles_y);
    item /= tiles_y;
    const int lvl = lvl0 + (int)(item % n_lvl), b = (int)(item / n_lvl);
    const int H = c.h >> lvl, W = c.w >> lvl;
    const int PS = 2 * c.radius + 2;  // side of the integer patch under a (2 r + 1)^2 window of bilinear samples
    if (tid < kTQ) {
        const int qy = ty * kTQH + tid / kTQW, qx = tx * kTQW + tid % kTQW;
        const bool ok = qy < c.h && qx < c.w;
        int pix = -1, x0 = 0, y0 = 0;
        float fx = 0.f, fy = 0.f;
        if (ok) {
            pix = qy * c.w + qx;
            const Patch pt = patch_of(c, coords, b, pix, lvl);
            x0 = pt.x0; y0 = pt.y0; fx = pt.fx; fy = pt.fy;
        }
        q_pix[tid] = pix; q_x0[tid] = x0; q_y0[tid] = y0; q_fx[tid] = fx; q_fy[tid] = fy;
        // bounding box of the patches, clipped to the map (rows outside are zero and are never staged): wave 0, lanes 0-31
        // (a patch that lies outside the map altogether contributes zeros and no rows)
        const bool some = ok && x0 < W && y0 < H && x0 > -PS && y0 > -PS;
        int lo_x = some ? max(x0, 0) : (1 << 30), lo_y = some ? max(y0, 0) : (1 << 30);
        int hi_x = some ? min(x0 + PS, W) : -(1 << 30), hi_y = some ? min(y0 + PS, H) : -(1 << 30);
#pragma unroll
        for (int m = kTQ / 2; m >= 1; m >>= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, m));
            lo_y = min(lo_y, __shfl_xor(lo_y, m));
            hi_x = max(hi_x, __shfl_xor(hi_x, m));
            hi_y = max(hi_y, __shfl_xor(hi_y, m));
        }
        if (tid == 0) {
            reg[0] = lo_x; reg[1] = lo_y;
            reg[2] = hi_x > lo_x ? hi_x - lo_x : 0;
            reg[3] = hi_y > lo_y ? hi_y - lo_y : 0;
        }
    }
    __syncthreads();
    const int rx0 = reg[0], ry0 = reg[1], rw = reg[2], rh = reg[3];
    const long R = (long)rw * rh;
    const int C = c.levels * (PS - 1) * (PS - 1);
    if (R > kTRMax) {  // per-query path (block-uniform branch)
        float (*Tw)[kTransposeStride] = reinterpret_cast<float (*)[kTransposeStride]>(lds + (size_t)wave * 64 * kTransposeStride * sizeof(float));
        for (int q = wave; q < kTQ; q += kWavesPerBlock) {
            const int pix = q_pix[q];
            if (pix >= 0) lookup_one<VEC>(c, fmap1, lp, coords, out, b, pix, lvl, lane, Pf[wave], Tw);
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    const int W7 = PS - 1;
    if (R == 0) {  // every patch lies outside the map: zeros
        for (int idx = tid; idx < kTQ * W7 * W7; idx += 256) {
            const int q = idx / (W7 * W7), tap = idx - q * (W7 * W7);
            if (q_pix[q] >= 0) out[((size_t)b * hw + q_pix[q]) * C + lvl * W7 * W7 + tap] = 0.f;
        }
        return;
    }
    unsigned char* A_hi = lds;
    unsigned char* A_lo = lds + kTLdsA / 2;
    unsigned char* B_hi = lds + kTLdsA;
    unsigned char* B_lo = B_hi + kTLdsB / 2;
    const float* f2 = lp.f2[lvl] + (size_t)b * H * W * D;
    const int Rn = (int)R, n_tiles = (Rn + 31) / 32;
    // per-thread row offsets of the region's 16-B chunks (one K chunk = 16 chunks of 4 floats per row): the same for every K chunk
    constexpr int CPR = kTKC / 4;                       // 16
    constexpr int NB = kTRMax * CPR / 256;              // 16 chunks per thread
    int b_off[NB];
    const float inv_rw = 1.0f / (float)rw;
#pragma unroll
    for (int u = 0; u < NB; u++) {
        const int i = tid + u * 256, p = i / CPR, c4 = i % CPR;
        int py = (int)(((float)p + 0.5f) * inv_rw);  // p / rw for p < 256 (a float reciprocal, corrected: no integer division per chunk)
        py -= (py * rw > p) ? 1 : 0;
        py += ((py + 1) * rw <= p) ? 1 : 0;
        const int px = p - py * rw;
        b_off[u] = p < Rn ? ((ry0 + py) * W + rx0 + px) * D + c4 * 4 : -1;
    }
    int a_off[2 * MT];
#pragma unroll
    for (int u = 0; u < 2 * MT; u++) {
        const int i = tid + u * 256, q = i / CPR, c4 = i % CPR;
        a_off[u] = q_pix[q] >= 0 ? q_pix[q] * D + c4 * 4 : -1;
    }
    const float* f1 = fmap1 + (size_t)b * hw * D;
    const int r = lane & 31, h = lane >> 5;
    cf16v acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][j][e] = 0.f;
    // (prefetching chunk k + 1 under chunk k needs a second register set: 288 registers, one block per CU -- measured 103 vs 66 us)
    constexpr int NCH = D / kTKC;
    float4 va[1][2 * MT], vb[1][NB];
    auto load_chunk = [&](int kc, float4 (&xa)[2 * MT], float4 (&xb)[NB]) {
#pragma unroll
        for (int u = 0; u < 2 * MT; u++) xa[u] = a_off[u] >= 0 ? *reinterpret_cast<const float4*>(f1 + a_off[u] + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < NB; u++)
            xb[u] = b_off[u] >= 0 ? *reinterpret_cast<const float4*>(f2 + b_off[u] + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        float4 (&xa)[2 * MT] = va[0];
        float4 (&xb)[NB] = vb[0];
        load_chunk(ch * kTKC, xa, xb);
        if (ch > 0) __syncthreads();  // the previous chunk's fragments have been read
#pragma unroll
        for (int u = 0; u < 2 * MT; u++) {
            const int i = tid + u * 256, q = i / CPR, c4 = i % CPR;
            uint2 hi, lo;
            corr_split4(xa[u], &hi, &lo);
            const int o = (c4 >> 1) * AS + q * 16 + (c4 & 1) * 8;
            *reinterpret_cast<uint2*>(A_hi + o) = hi;
            *reinterpret_cast<uint2*>(A_lo + o) = lo;
        }
#pragma unroll
        for (int u = 0; u < NB; u++) {
            const int i = tid + u * 256, p = i / CPR, c4 = i % CPR;
            if (p < n_tiles * 32) {  // (rows of the last 32-row tile beyond the region: zeros)
                uint2 hi, lo;
                corr_split4(xb[u], &hi, &lo);
                const int o = (c4 >> 1) * BS + p * 16 + (c4 & 1) * 8;
                *reinterpret_cast<uint2*>(B_hi + o) = hi;
                *reinterpret_cast<uint2*>(B_lo + o) = lo;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTKC / 16; kk++) {
            const int k8 = kk * 2 + h;
            uint4 ah[MT], al[MT];
#pragma unroll
            for (int m = 0; m < MT; m++) {
                ah[m] = *reinterpret_cast<const uint4*>(A_hi + k8 * AS + (m * 32 + r) * 16);
                al[m] = *reinterpret_cast<const uint4*>(A_lo + k8 * AS + (m * 32 + r) * 16);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int nt = wave * 2 + j;
                if (nt < n_tiles) {
                    const uint4 bh = *reinterpret_cast<const uint4*>(B_hi + k8 * BS + (nt * 32 + r) * 16);
                    const uint4 bl = *reinterpret_cast<const uint4*>(B_lo + k8 * BS + (nt * 32 + r) * 16);
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf8*>(&al[m]), *reinterpret_cast<const cbf8*>(&bh), acc[m][j], 0, 0, 0);
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf8*>(&ah[m]), *reinterpret_cast<const cbf8*>(&bl), acc[m][j], 0, 0, 0);
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf8*>(&ah[m]), *reinterpret_cast<const cbf8*>(&bh), acc[m][j], 0, 0, 0);
                    }
                }
            }
        }
    }
    __syncthreads();  // every fragment read is done: C takes the place of the operand planes
    float* Cm = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int nt = wave * 2 + j;
        if (nt < n_tiles) {
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int q = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;  // row of the 32 x 32 result held by register e
                    Cm[q * kTCStride + nt * 32 + r] = acc[m][j][e];
                }
        }
    }
    __syncthreads();
    const float inv = 1.0f / sqrtf((float)D);  // corr.py:56
    for (int idx = tid; idx < kTQ * W7 * W7; idx += 256) {
        const int q = idx / (W7 * W7), tap = idx - q * (W7 * W7);
        const int pix = q_pix[q];
        if (pix < 0) continue;
        // consecutive lanes walk a window ROW (consecutive LDS addresses); the output index is a * W7 + bb (a: x offset index, bb: y offset
        // index, corr.py:31-35 order)
        const int bb = tap / W7, a = tap - bb * W7;
        const int lx = q_x0[q] - rx0 + a, ly = q_y0[q] - ry0 + bb;
        const float* row = Cm + q * kTCStride;
        auto at = [&](int x, int y) { return ((unsigned)x < (unsigned)rw && (unsigned)y < (unsigned)rh) ? row[y * rw + x] : 0.f; };
        const float p00 = at(lx, ly), p10 = at(lx + 1, ly), p01 = at(lx, ly + 1), p11 = at(lx + 1, ly + 1);
        const float fx = q_fx[q], fy = q_fy[q];
        const float v = (1.f - fx) * (1.f - fy) * p00 + fx * (1.f - fy) * p10 + (1.f - fx) * fy * p01 + fx * fy * p11;
        out[((size_t)b * hw + pix) * C + lvl * W7 * W7 + a * W7 + bb] = v * inv;
    }
}

// ONE launch: blocks [0, fine) walk `n_fine` levels with 4 x 8-query tiles, blocks [fine, fine + wide) the remaining levels with 8 x 8-query
// tiles (both ranges multiples of 8 and dealt XCD-aware: every XCD gets a contiguous range of neighbouring tiles of one level)
template <int VEC>
__global__ __launch_bounds__(256, 2) void corr_lookup_tiled_kernel(liso_corr_cfg c, const float* __restrict__ fmap1, LevelPtrs lp,
                                                                const float* __restrict__ coords, float* __restrict__ out, int tiles_x,
                                                                int tiles_y4, int tiles_y8, int n_fine, unsigned fine_blocks, unsigned wide_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (blockIdx.x < fine_blocks) {
        const long per_xcd = fine_blocks / 8;
        const long item = (long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
        if (item < (long)c.batch * n_fine * tiles_y4 * tiles_x)
            corr_tile_body<VEC, 1, 256>(c, fmap1, lp, coords, out, tiles_x, tiles_y4, 0, n_fine, item, lds);
    } else {
        const unsigned bid = blockIdx.x - fine_blocks;
        const long per_xcd = wide_blocks / 8;
        const long item = (long)(bid % 8) * per_xcd + bid / 8;
        if (item < (long)c.batch * (c.levels - n_fine) * tiles_y8 * tiles_x)
            corr_tile_body<VEC, 2, 192>(c, fmap1, lp, coords, out, tiles_x, tiles_y8, n_fine, c.levels - n_fine, item, lds);
    }
}

// Backward of the lookup, step 1: the adjoint of (bilinear weights x 7x7 window) for one (query pixel, level) is an 8x8
// patch of coefficients gP[u][v] on the integer grid of that level.  They are ADDED into the dense per-level matrix
//     dvol_i[b, p, y*W_i + x]    ( == d loss / d (pooled correlation volume of level i) )
// One lane per patch entry; each (p, x, y) is owned by exactly one lane of one launch, so a plain read-modify-write is
// race free and the 6 RAFT iterations of one direction accumulate into the same matrices launch after launch.
// Step 2 (host, once per direction): grad_fmap1 = sum_i dvol_i @ f2_i, grad_f2_i = dvol_i^T @ fmap1 -- plain GEMMs.
__global__ __launch_bounds__(64 * kWavesPerBlock) void corr_lookup_bwd_dvol_kernel(liso_corr_cfg c, const float* __restrict__ coords,
                                                                                   const float* __restrict__ grad_out,
                                                                                   LevelPtrs lp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hw = c.h * c.w;
    const long item = (long)blockIdx.x * kWavesPerBlock + wave;
    const long total = (long)c.batch * hw * c.levels;
    if (item >= total) return;
    const int lvl = (int)(item % c.levels);
    const int pix = (int)((item / c.levels) % hw);
    const int b = (int)(item / ((long)c.levels * hw));
    const Patch pt = patch_of(c, coords, b, pix, lvl);
    const int W7 = 2 * c.radius + 1, C = c.levels * W7 * W7;
    const int u = lane & 7, v = lane >> 3;
    const int x = pt.x0 + u, y = pt.y0 + v;
    if (x < 0 || x >= pt.W || y < 0 || y >= pt.H) return;  // zeros padding of grid_sample: no gradient outside
    const float* go = grad_out + ((size_t)b * hw + pix) * C + lvl * W7 * W7;
    float s = 0.f;
    if (u < W7 && v < W7) s += (1.f - pt.fx) * (1.f - pt.fy) * go[u * W7 + v];
    if (u >= 1 && u - 1 < W7 && v < W7) s += pt.fx * (1.f - pt.fy) * go[(u - 1) * W7 + v];
    if (u < W7 && v >= 1 && v - 1 < W7) s += (1.f - pt.fx) * pt.fy * go[u * W7 + v - 1];
    if (u >= 1 && u - 1 < W7 && v >= 1 && v - 1 < W7) s += pt.fx * pt.fy * go[(u - 1) * W7 + v - 1];
    if (s == 0.f) return;
    float* dst = lp.g2[lvl] + ((size_t)b * hw + pix) * ((size_t)pt.H * pt.W) + (size_t)y * pt.W + x;
    *dst += s * (1.0f / sqrtf((float)c.dim));
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

inline bool cfg_ok(const liso_corr_cfg* c) {
    return c && c->batch >= 1 && c->h >= 1 && c->w >= 1 && (c->dim == 128 || c->dim == 256) && c->levels >= 1 &&
           c->levels <= LISO_CORR_MAX_LEVELS && c->radius >= 0 && c->radius <= 3 && (c->h >> (c->levels - 1)) >= 1 &&
           (c->w >> (c->levels - 1)) >= 1;
}

}  // namespace

extern "C" {

int liso_corr_lookup_fwd_f32(const liso_corr_cfg* cfg, const float* fmap1, const float* const* fmap2_levels,
                             const float* coords, float* out, void* stream) {
    if (!cfg_ok(cfg) || !fmap1 || !fmap2_levels || !coords || !out) return LISO_EINVAL;
    LevelPtrs lp = {};
    for (int i = 0; i < cfg->levels; i++) {
        if (!fmap2_levels[i]) return LISO_EINVAL;
        lp.f2[i] = fmap2_levels[i];
    }
    const long total = (long)cfg->batch * cfg->h * cfg->w * cfg->levels;
    const unsigned grid = (unsigned)(((total + kWavesPerBlock - 1) / kWavesPerBlock + 7) / 8 * 8);  // XCD-aware order
    hipStream_t st = (hipStream_t)stream;
    if (cfg->dim == 128)
        corr_lookup_fwd_kernel<1><<<grid, 64 * kWavesPerBlock, 0, st>>>(*cfg, fmap1, lp, coords, out);
    else
        corr_lookup_fwd_kernel<2><<<grid, 64 * kWavesPerBlock, 0, st>>>(*cfg, fmap1, lp, coords, out);
    return check_launch();
}

int liso_corr_lookup_fwd_tiled_f32(const liso_corr_cfg* cfg, const float* fmap1, const float* const* fmap2_levels, const float* coords,
                                   float* out, void* stream) {
    if (!cfg_ok(cfg) || !fmap1 || !fmap2_levels || !coords || !out) return LISO_EINVAL;
    if ((((uintptr_t)fmap1) & 15) != 0) return LISO_EINVAL;
    LevelPtrs lp = {};
    for (int i = 0; i < cfg->levels; i++) {
        if (!fmap2_levels[i] || (((uintptr_t)fmap2_levels[i]) & 15) != 0) return LISO_EINVAL;
        lp.f2[i] = fmap2_levels[i];
    }
    hipStream_t st = (hipStream_t)stream;
    const int tiles_x = (cfg->w + kTQW - 1) / kTQW, tiles_y4 = (cfg->h + 3) / 4, tiles_y8 = (cfg->h + 7) / 8;
    constexpr int kLds = 2 * (kTKC / 8) * ((32 + 256) * 16 + 128);  // (the 8 x 8-query shape needs less: 2 * 8 * ((64 + 192) * 16 + 128) - 8 KB ... both <= this)
    static liso_dev::PerDeviceFlag attr_set1, attr_set2;
    if (!liso_dev::lds_opt_in(attr_set1, (const void*)corr_lookup_tiled_kernel<1>, kLds) ||
        !liso_dev::lds_opt_in(attr_set2, (const void*)corr_lookup_tiled_kernel<2>, kLds))
        return LISO_ELAUNCH;
    // 8 x 8-query tiles on the pooled levels once they give every CU a block (4 / 2 samples of 64 x 64 queries: 768 / 384 such blocks,
    // 49.8 vs 65.5 / 32.8 vs 34.5 us; one sample: 192 blocks, no gain); LISO_CORR_WIDE_TILES = 0 / 1 forces either
    static const int wide_env = getenv("LISO_CORR_WIDE_TILES") ? atoi(getenv("LISO_CORR_WIDE_TILES")) : -1;
    const long wide_all = (long)cfg->batch * (cfg->levels - 1) * tiles_y8 * tiles_x;
    const bool wide = cfg->levels > 1 && (wide_env < 0 ? wide_all >= 256 : wide_env != 0);
    const int n_fine = wide ? 1 : cfg->levels;
    const unsigned fine_blocks = (unsigned)(((long)cfg->batch * n_fine * tiles_y4 * tiles_x + 7) / 8 * 8);
    const unsigned wide_blocks = wide ? (unsigned)((wide_all + 7) / 8 * 8) : 0u;
    if (cfg->dim == 128)
        corr_lookup_tiled_kernel<1><<<fine_blocks + wide_blocks, 256, kLds, st>>>(*cfg, fmap1, lp, coords, out, tiles_x, tiles_y4, tiles_y8, n_fine,
                                                                                fine_blocks, wide_blocks);
    else
        corr_lookup_tiled_kernel<2><<<fine_blocks + wide_blocks, 256, kLds, st>>>(*cfg, fmap1, lp, coords, out, tiles_x, tiles_y4, tiles_y8, n_fine,
                                                                                fine_blocks, wide_blocks);
    return check_launch();
}

int liso_corr_lookup_bwd_dvol_f32(const liso_corr_cfg* cfg, const float* coords, const float* grad_out,
                                  float* const* dvol_levels, void* stream) {
    if (!cfg_ok(cfg) || !coords || !grad_out || !dvol_levels) return LISO_EINVAL;
    LevelPtrs lp = {};
    for (int i = 0; i < cfg->levels; i++) {
        if (!dvol_levels[i]) return LISO_EINVAL;
        lp.g2[i] = dvol_levels[i];
    }
    const long total = (long)cfg->batch * cfg->h * cfg->w * cfg->levels;
    const unsigned grid = (unsigned)((total + kWavesPerBlock - 1) / kWavesPerBlock);
    corr_lookup_bwd_dvol_kernel<<<grid, 64 * kWavesPerBlock, 0, (hipStream_t)stream>>>(*cfg, coords, grad_out, lp);
    return check_launch();
}

}  // extern "C"
