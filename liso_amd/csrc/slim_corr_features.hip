// The FEATURE side of the RAFT correlation for gfx950 (MI355X).  C ABI + reference lines: include/liso_slim.h.
//   (1) the pooled fmap2 pyramid, forward and backward, one launch each (the reference pools the correlation VOLUME, corr.py:20-21;
//       pooling is linear, so pooling the features gives the same lookups -- slim_corr.hip);
//   (2) backward of the lookup, step 2 of 2: the two dense contractions that turn the volume gradients dvol_l
//       (liso_corr_lookup_bwd_dvol_f32, slim_corr.hip) into feature gradients:
//
//     grad_fmap1[b] (hw x D)    = sum_l  dvol_l[b] (hw x HW_l)   . f2_l[b]  (HW_l x D)        "G1": one product over the concatenated levels
//     grad_f2_l[b]  (HW_l x D)  =        dvol_l[b]^T (HW_l x hw) . fmap1[b] (hw x D)          "G2": all levels in one launch
//
// (the adjoint of corr = fmap1^T fmap2 / sqrt(D), liso/slim/model/raft_code/corr.py:48-56, pooled per level :20-21; the 1 / sqrt(D) is
// already inside dvol).  Until round 5 these were eight rocBLAS batched GEMMs per training step (fp32 MFMA, 0.26 ms); here they are two
// launches (+ two fixed-order reductions of the split-K partial sums) in the arithmetic of the convolutions around them: F32X3 (fp32
// operands split into bf16 hi / lo on the way into LDS, hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_bf16) or exact fp32
// (v_mfma_f32_32x32x2_f32) in the parity configuration.
//
// Every operand that has its reduction index as the ROW index in memory -- f2_l and fmap1 ([k][channel]) always, dvol in G2 ([query][cell]
// read as [k = query][m = cell]) -- is staged into LDS exactly as it lies and read back with ds_read_b64_tr_b16, gfx950's transposing LDS
// read (the idiom of conv_wgrad.hip); dvol in G1 ([m = query][k = cell]) is k-contiguous and takes the [k8][row][8] layout with plain
// 16-byte fragment reads.  Block = 256 threads = 2 x 2 waves on a 128 x 128 tile, K slabs of 32, the next slab's global loads in
// registers under the MFMAs of the current one.  Blocks that share a (sample, K range) -- and with it the B panel they stream from L2 --
// are dealt to the same XCD.  K is split so that the launch fills the chip; partial tiles go to a workspace and are added in split
// order by a second kernel: no float atomics, bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/liso_conv.h"
#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim.h"

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef s4 __attribute__((address_space(3))) * lds_s4_ptr;

constexpr int kBM = 128, kBN = 128, kBK = 32, kThreads = 256;
constexpr int kMaxSplits = 16;
// F32X3 planes: [k][128 bf16 + 64 B] (the four rows a transposing read touches tile the 64 banks) and, for the k-contiguous A of G1,
// [k8][128 rows][8 bf16] + 64 B per k8 group (the eight groups a wave's split stores touch start 64 B apart)
constexpr int kPS = kBN * 2 + 64;         // 320
constexpr int kAS = kBM * 16 + 64;        // 2112
constexpr int kPlaneT = kBK * kPS;        // 10240: one transposed-read plane (A of G2, B)
constexpr int kPlaneN = (kBK / 8) * kAS;  // 8448:  one plane of the k-contiguous A
// exact fp32: [k][160 floats] (two half-waves read rows 2 s and 2 s + 1: 160 mod 64 = 32 banks apart), A of G1 [row][33 floats]
constexpr int kPF = 160, kAF = 33;

struct BwdArgs {
    const float* dvol[LISO_CORR_MAX_LEVELS];
    const float* f2[LISO_CORR_MAX_LEVELS];
    float* g2[LISO_CORR_MAX_LEVELS];
    const float* fmap1;
    float* g1;
    float* part;                              // split-K partial tiles (splits > 1)
    int hw, dim, levels, batch, splits;
    int cells[LISO_CORR_MAX_LEVELS];          // HW_l
    int slab0[LISO_CORR_MAX_LEVELS + 1];      // G1: first K slab of level l in the concatenated reduction
    int mt0[LISO_CORR_MAX_LEVELS + 1];        // G2: first M tile of level l
    int row0[LISO_CORR_MAX_LEVELS + 1];       // G2: first row of level l in the partial buffer (sum of HW_l)
    int m_tiles;                              // G1: ceil(hw / 128); G2: mt0[levels]
    int n_tiles;                              // dim / 128
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ void split4(const float4 v, uint2* hi, uint2* lo) {
    const unsigned p0 = pack_bf16(v.x, v.y), p1 = pack_bf16(v.z, v.w);
    *hi = make_uint2(p0, p1);
    *lo = make_uint2(pack_bf16(v.x - __uint_as_float(p0 << 16), v.y - __uint_as_float(p0 & 0xffff0000u)),
                     pack_bf16(v.z - __uint_as_float(p1 << 16), v.w - __uint_as_float(p1 & 0xffff0000u)));
}
__device__ __forceinline__ bf8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
    const s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)p0);
    const s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)p1);
    typedef short s8 __attribute__((ext_vector_type(8)));
    const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf8, v);
}

// 4 consecutive floats at p, of which the first n (0 .. 4) exist; VEC: n is 0 or 4 and p is 16-byte aligned
template <bool VEC>
__device__ __forceinline__ float4 load4(const float* __restrict__ p, int n) {
    if constexpr (VEC) {
        return n > 0 ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n > 0) v.x = p[0];
        if (n > 1) v.y = p[1];
        if (n > 2) v.z = p[2];
        if (n > 3) v.w = p[3];
        return v;
    }
}

// One K slab of one operand in registers: 4 float4 per thread.
//   k-contiguous (A of G1):  rows m = (tid >> 3) + 32 u, floats k4 = (tid & 7) * 4 .. + 3 of the slab     (128 rows x 32 k)
//   row = k  (everything else): rows k = (tid >> 5) + 8 u, floats c4 = (tid & 31) * 4 .. + 3 of the tile  (32 k x 128 columns)
struct Slab {
    float4 v[4];
};

template <int MODE, bool TA, bool VEC>
__global__ __launch_bounds__(kThreads, 2) void corr_bwd_gemm_kernel(const BwdArgs a) {
    constexpr bool X3 = MODE == LISO_CONV_F32X3;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> (group = (sample, split, n tile), m tile): the groups' id is the FAST index, so that blocks dealt round-robin to the XCDs
    // with the same (sample, K range) -- the same B panel -- meet in one L2
    const int groups = a.batch * a.splits * a.n_tiles;
    int gidx = blockIdx.x % groups;
    const int mt = blockIdx.x / groups;
    const int nt = gidx % a.n_tiles;
    gidx /= a.n_tiles;
    const int split = gidx % a.splits, b = gidx / a.splits;
    const int n0 = nt * kBN;
    const int D = a.dim, hw = a.hw;

    // ---- what this block multiplies ------------------------------------------------------------------------------------------------
    int lvl = 0, m0, M, s_begin, s_end;
    if constexpr (TA) {  // G2: one level, M = its cells, K = hw
        while (lvl + 1 < a.levels && mt >= a.mt0[lvl + 1]) lvl++;
        m0 = (mt - a.mt0[lvl]) * kBM;
        M = a.cells[lvl];
        const int slabs = (hw + kBK - 1) / kBK;
        s_begin = (int)((long)slabs * split / a.splits);
        s_end = (int)((long)slabs * (split + 1) / a.splits);
    } else {  // G1: M = hw, K = the levels' cells one after the other
        m0 = mt * kBM;
        M = hw;
        const int slabs = a.slab0[a.levels];
        s_begin = (int)((long)slabs * split / a.splits);
        s_end = (int)((long)slabs * (split + 1) / a.splits);
    }

    unsigned char* As = lds;
    unsigned char* Bs;
    if constexpr (X3) Bs = lds + 2 * (TA ? kPlaneT : kPlaneN);
    else Bs = lds + (TA ? kBK * kPF * 4 : kBM * kAF * 4);

    auto load_slab = [&](int s, Slab& xa, Slab& xb) {
        const float *Ap, *Bp;
        int K, k0, lda;
        if constexpr (TA) {
            K = hw; k0 = s * kBK; lda = a.cells[lvl];
            Ap = a.dvol[lvl] + (size_t)b * hw * lda;
            Bp = a.fmap1 + (size_t)b * hw * D;
        } else {
            int l = 0;
            while (l + 1 < a.levels && s >= a.slab0[l + 1]) l++;
            K = a.cells[l]; k0 = (s - a.slab0[l]) * kBK; lda = K;
            Ap = a.dvol[l] + (size_t)b * hw * lda;
            Bp = a.f2[l] + (size_t)b * K * D;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if constexpr (TA) {
                const int k = k0 + (tid >> 5) + 8 * u, m = m0 + (tid & 31) * 4;
                const int n = k < K ? min(max(M - m, 0), 4) : 0;
                xa.v[u] = load4<VEC>(Ap + (size_t)(k < K ? k : 0) * lda + (m < M ? m : 0), n);
            } else {
                const int m = m0 + (tid >> 3) + 32 * u, k = k0 + (tid & 7) * 4;
                const int n = m < M ? min(max(K - k, 0), 4) : 0;
                xa.v[u] = load4<VEC>(Ap + (size_t)(m < M ? m : 0) * lda + (k < K ? k : 0), n);
            }
            const int kb = k0 + (tid >> 5) + 8 * u;
            xb.v[u] = load4<true>(Bp + (size_t)(kb < K ? kb : 0) * D + n0 + (tid & 31) * 4, kb < K ? 4 : 0);  // (D % 128 == 0: whole, aligned)
        }
    };
    auto store_slab = [&](const Slab& xa, const Slab& xb) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kb = (tid >> 5) + 8 * u, c4 = (tid & 31) * 4;
            if constexpr (X3) {
                uint2 hi, lo;
                split4(xb.v[u], &hi, &lo);
                *reinterpret_cast<uint2*>(Bs + kb * kPS + c4 * 2) = hi;
                *reinterpret_cast<uint2*>(Bs + kPlaneT + kb * kPS + c4 * 2) = lo;
                split4(xa.v[u], &hi, &lo);
                if constexpr (TA) {
                    *reinterpret_cast<uint2*>(As + kb * kPS + c4 * 2) = hi;
                    *reinterpret_cast<uint2*>(As + kPlaneT + kb * kPS + c4 * 2) = lo;
                } else {
                    const int m = (tid >> 3) + 32 * u, k4 = (tid & 7) * 4;
                    const int o = (k4 >> 3) * kAS + m * 16 + (k4 & 4) * 2;
                    *reinterpret_cast<uint2*>(As + o) = hi;
                    *reinterpret_cast<uint2*>(As + kPlaneN + o) = lo;
                }
            } else {
                *reinterpret_cast<float4*>(Bs + (kb * kPF + c4) * 4) = xb.v[u];
                if constexpr (TA) {
                    *reinterpret_cast<float4*>(As + (kb * kPF + c4) * 4) = xa.v[u];
                } else {
                    const int m = (tid >> 3) + 32 * u, k4 = (tid & 7) * 4;
                    float* row = reinterpret_cast<float*>(As) + m * kAF + k4;
                    row[0] = xa.v[u].x; row[1] = xa.v[u].y; row[2] = xa.v[u].z; row[3] = xa.v[u].w;
                }
            }
        }
    };

    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;          // transposing-read lane geometry (16-lane groups)
    const int tr_lane = (8 * (g >> 1) + q) * kPS + (16 * (g & 1) + 4 * p) * 2;

    auto mfma_slab = [&]() {
        if constexpr (X3) {
#pragma unroll
            for (int kk = 0; kk < kBK / 16; kk++) {
                bf8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const unsigned char* bp = Bs + kk * 16 * kPS + tr_lane + (wn * 64 + j * 32) * 2;
                    bh[j] = tr_pair(bp, bp + 4 * kPS);
                    bl[j] = tr_pair(bp + kPlaneT, bp + kPlaneT + 4 * kPS);
                }
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    if constexpr (TA) {
                        const unsigned char* ap = As + kk * 16 * kPS + tr_lane + (wm * 64 + i * 32) * 2;
                        ah[i] = tr_pair(ap, ap + 4 * kPS);
                        al[i] = tr_pair(ap + kPlaneT, ap + kPlaneT + 4 * kPS);
                    } else {
                        const unsigned char* ap = As + (kk * 2 + h) * kAS + (wm * 64 + i * 32 + r) * 16;
                        ah[i] = *reinterpret_cast<const bf8*>(ap);
                        al[i] = *reinterpret_cast<const bf8*>(ap + kPlaneN);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
            const float* Af = reinterpret_cast<const float*>(As);
            const float* Bf = reinterpret_cast<const float*>(Bs);
#pragma unroll 4
            for (int s2 = 0; s2 < kBK / 2; s2++) {
                const int k = 2 * s2 + h;
                float av[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    const int row = wm * 64 + i * 32 + r;
                    av[i] = TA ? Af[k * kPF + row] : Af[row * kAF + k];
                }
#pragma unroll
                for (int j = 0; j < 2; j++) bv[j] = Bf[k * kPF + wn * 64 + j * 32 + r];
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    if (s_begin < s_end) {
        Slab xa, xb;
        load_slab(s_begin, xa, xb);
        store_slab(xa, xb);
        __syncthreads();
        for (int s = s_begin; s < s_end; s++) {
            const bool more = s + 1 < s_end;
            load_slab(more ? s + 1 : s, xa, xb);  // (the last pass re-reads its own slab: no branch around the loads)
            mfma_slab();
            __syncthreads();
            if (more) {
                store_slab(xa, xb);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: register e of a 32 x 32 result = row (e & 3) + 8 (e >> 2) + 4 h, column r -----------------------------------------
    float* dst;
    size_t ld = (size_t)D;
    if (a.splits > 1) {
        if constexpr (TA) dst = a.part + ((size_t)(split * a.batch + b) * a.row0[a.levels] + a.row0[lvl]) * D;
        else dst = a.part + (size_t)(split * a.batch + b) * hw * D;
    } else {
        if constexpr (TA) dst = a.g2[lvl] + (size_t)b * M * D;
        else dst = a.g1 + (size_t)b * hw * D;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < M) dst[(size_t)row * ld + n0 + wn * 64 + j * 32 + r] = acc[i][j][e];
            }
}

// out = sum over the splits, in split order (fixed: bitwise reproducible).  G1: out = g1 [B][hw][D].  G2: rows of the partial buffer
// [B][sum HW_l][D] go to the levels' own tensors.
__global__ __launch_bounds__(256) void corr_bwd_reduce_kernel(const BwdArgs a, int is_g2) {
    const int D4 = a.dim / 4;
    const long rows_per_sample = is_g2 ? a.row0[a.levels] : a.hw;
    const long total = (long)a.batch * rows_per_sample * D4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float4* p = reinterpret_cast<const float4*>(a.part) + i;
    float4 s = p[0];
    for (int k = 1; k < a.splits; k++) {
        const float4 v = p[(size_t)k * total];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long row = i / D4;
    const int c4 = (int)(i - row * D4);
    if (!is_g2) {
        reinterpret_cast<float4*>(a.g1)[i] = s;
        return;
    }
    const int b = (int)(row / rows_per_sample), rr = (int)(row - (long)b * rows_per_sample);
    int l = 0;
    while (l + 1 < a.levels && rr >= a.row0[l + 1]) l++;
    reinterpret_cast<float4*>(a.g2[l])[((size_t)b * a.cells[l] + rr - a.row0[l]) * D4 + c4] = s;
}


// ---- pooled pyramid ---------------------------------------------------------------------------------------------------------------
// level l = avg_pool2d(level l - 1, 2, stride 2) (floor: an odd last row / column is dropped, as F.avg_pool2d does), channels last.
// One thread owns 4 channels of one 8 x 8 block of level-0 pixels and builds every level's pixels inside it in registers: a level-l
// pixel exists iff its index is < H_l = H >> l, and then all four pixels under it exist.  Sums in (y, x), (y, x + 1), (y + 1, x),
// (y + 1, x + 1) order, times 0.25.
struct PyrPtrs {
    float* lvl[LISO_CORR_MAX_LEVELS];
    const float* grad[LISO_CORR_MAX_LEVELS];
};

__device__ __forceinline__ float4 mean4(const float4 a, const float4 b, const float4 c, const float4 d) {
    return make_float4((((a.x + b.x) + c.x) + d.x) * 0.25f, (((a.y + b.y) + c.y) + d.y) * 0.25f, (((a.z + b.z) + c.z) + d.z) * 0.25f,
                       (((a.w + b.w) + c.w) + d.w) * 0.25f);
}

__global__ __launch_bounds__(256) void corr_pyramid_fwd_kernel(liso_corr_cfg c, const float* __restrict__ f2, PyrPtrs pp) {
    const int D4 = c.dim / 4;
    const int bw = (c.w + 7) / 8, bh = (c.h + 7) / 8;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)c.batch * bh * bw * D4) return;
    const int c4 = (int)(i % D4);
    i /= D4;
    const int bx = (int)(i % bw);
    i /= bw;
    const int by = (int)(i % bh), b = (int)(i / bh);
    const float4* src = reinterpret_cast<const float4*>(f2) + (size_t)b * c.h * c.w * D4 + c4;
    const int H1 = c.h >> 1, W1 = c.w >> 1, H2 = c.h >> 2, W2 = c.w >> 2, H3 = c.h >> 3, W3 = c.w >> 3;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 l2[2][2];
#pragma unroll
    for (int qy = 0; qy < 2; qy++)
#pragma unroll
        for (int qx = 0; qx < 2; qx++) {  // one level-2 pixel = 2 x 2 level-1 pixels = 4 x 4 level-0 pixels
            float4 l1[2][2];
#pragma unroll
            for (int ry = 0; ry < 2; ry++)
#pragma unroll
                for (int rx = 0; rx < 2; rx++) {
                    const int y1 = by * 4 + qy * 2 + ry, x1 = bx * 4 + qx * 2 + rx;
                    float4 m = zero;
                    if (y1 < H1 && x1 < W1 && c.levels > 1) {
                        const float4* p = src + ((size_t)(2 * y1) * c.w + 2 * x1) * D4;
                        m = mean4(p[0], p[D4], p[(size_t)c.w * D4], p[(size_t)c.w * D4 + D4]);
                        reinterpret_cast<float4*>(pp.lvl[1])[((size_t)b * H1 * W1 + (size_t)y1 * W1 + x1) * D4 + c4] = m;
                    }
                    l1[ry][rx] = m;
                }
            const int y2 = by * 2 + qy, x2 = bx * 2 + qx;
            float4 m2 = zero;
            if (y2 < H2 && x2 < W2 && c.levels > 2) {
                m2 = mean4(l1[0][0], l1[0][1], l1[1][0], l1[1][1]);
                reinterpret_cast<float4*>(pp.lvl[2])[((size_t)b * H2 * W2 + (size_t)y2 * W2 + x2) * D4 + c4] = m2;
            }
            l2[qy][qx] = m2;
        }
    if (by < H3 && bx < W3 && c.levels > 3)
        reinterpret_cast<float4*>(pp.lvl[3])[((size_t)b * H3 * W3 + (size_t)by * W3 + bx) * D4 + c4] = mean4(l2[0][0], l2[0][1], l2[1][0], l2[1][1]);
}

// adjoint: grad wrt level 0 = g0 + 0.25 (g1 + 0.25 (g2 + 0.25 g3)) at the pixels above (where they exist); absent gradients (NULL) are zeros
__global__ __launch_bounds__(256) void corr_pyramid_bwd_kernel(liso_corr_cfg c, PyrPtrs pp, float* __restrict__ out) {
    const int D4 = c.dim / 4;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)c.batch * c.h * c.w * D4) return;
    const int c4 = (int)(i % D4);
    long pix = i / D4;
    const int x = (int)(pix % c.w);
    pix /= c.w;
    const int y = (int)(pix % c.h), b = (int)(pix / c.h);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = c.levels - 1; l >= 0; l--) {
        g = make_float4(g.x * 0.25f, g.y * 0.25f, g.z * 0.25f, g.w * 0.25f);
        const int Hl = c.h >> l, Wl = c.w >> l, yl = y >> l, xl = x >> l;
        // (floors nest: a level-l pixel above (y, x) exists iff yl < H_l, and then every level between exists above (y, x) too)
        if (yl >= Hl || xl >= Wl) {
            g = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        if (pp.grad[l]) {
            const float4 v = reinterpret_cast<const float4*>(pp.grad[l])[((size_t)b * Hl * Wl + (size_t)yl * Wl + xl) * D4 + c4];
            g = make_float4(g.x + v.x, g.y + v.y, g.z + v.z, g.w + v.w);
        }
    }
    reinterpret_cast<float4*>(out)[i] = g;
}

inline bool cfg_ok(const liso_corr_cfg* c) {
    return c && c->batch >= 1 && c->h >= 1 && c->w >= 1 && (c->dim == 128 || c->dim == 256) && c->levels >= 1 &&
           c->levels <= LISO_CORR_MAX_LEVELS && (c->h >> (c->levels - 1)) >= 1 && (c->w >> (c->levels - 1)) >= 1;
}

struct Plan {
    BwdArgs a;
    int splits1, splits2, m_tiles1, m_tiles2;
    bool vec;
};

inline int pick_splits(long tiles, int slabs) {
    static const int target = getenv("LISO_CORR_BWD_BLOCKS") ? atoi(getenv("LISO_CORR_BWD_BLOCKS")) : 512;  // (two blocks per CU: 143.7 us at B = 2, 64 x 64 against 171.5 at 256 and 226.4 at 128)
    int s = (int)((target + tiles - 1) / tiles);  // fill the CUs
    if (s > kMaxSplits) s = kMaxSplits;
    if (s > slabs) s = slabs;
    return s < 1 ? 1 : s;
}

inline Plan make_plan(const liso_corr_cfg* c) {
    Plan p = {};
    BwdArgs& a = p.a;
    a.hw = c->h * c->w; a.dim = c->dim; a.levels = c->levels; a.batch = c->batch;
    a.n_tiles = c->dim / kBN;
    p.vec = true;
    for (int l = 0; l < c->levels; l++) {
        a.cells[l] = (c->h >> l) * (c->w >> l);
        a.slab0[l + 1] = a.slab0[l] + (a.cells[l] + kBK - 1) / kBK;
        a.mt0[l + 1] = a.mt0[l] + (a.cells[l] + kBM - 1) / kBM;
        a.row0[l + 1] = a.row0[l] + a.cells[l];
        if (a.cells[l] % 4) p.vec = false;
    }
    p.m_tiles1 = (a.hw + kBM - 1) / kBM;
    p.m_tiles2 = a.mt0[c->levels];
    p.splits1 = pick_splits((long)p.m_tiles1 * c->batch * a.n_tiles, a.slab0[c->levels]);
    p.splits2 = pick_splits((long)p.m_tiles2 * c->batch * a.n_tiles, (a.hw + kBK - 1) / kBK);
    return p;
}

template <int MODE, bool TA>
int launch_gemm(const BwdArgs& a, bool vec, int m_tiles, hipStream_t st) {
    const unsigned grid = (unsigned)(m_tiles * a.batch * a.splits * a.n_tiles);
    size_t lds;
    if (MODE == LISO_CONV_F32X3) lds = 2 * (TA ? kPlaneT : kPlaneN) + 2 * kPlaneT;
    else lds = (TA ? kBK * kPF * 4 : kBM * kAF * 4) + kBK * kPF * 4;
    if (vec)
        corr_bwd_gemm_kernel<MODE, TA, true><<<grid, kThreads, lds, st>>>(a);
    else
        corr_bwd_gemm_kernel<MODE, TA, false><<<grid, kThreads, lds, st>>>(a);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // namespace

extern "C" {

int liso_corr_pyramid_fwd_f32(const liso_corr_cfg* cfg, const float* fmap2, float* const* levels, void* stream) {
    if (!cfg_ok(cfg) || !fmap2 || !levels || (((uintptr_t)fmap2) & 15) != 0) return LISO_EINVAL;
    PyrPtrs pp = {};
    for (int l = 1; l < cfg->levels; l++) {
        if (!levels[l] || (((uintptr_t)levels[l]) & 15) != 0) return LISO_EINVAL;
        pp.lvl[l] = levels[l];
    }
    if (cfg->levels == 1) return LISO_OK;
    const long total = (long)cfg->batch * ((cfg->h + 7) / 8) * ((cfg->w + 7) / 8) * (cfg->dim / 4);
    corr_pyramid_fwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(*cfg, fmap2, pp);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

int liso_corr_pyramid_bwd_f32(const liso_corr_cfg* cfg, const float* const* grad_levels, float* grad_fmap2, void* stream) {
    if (!cfg_ok(cfg) || !grad_levels || !grad_fmap2 || (((uintptr_t)grad_fmap2) & 15) != 0) return LISO_EINVAL;
    PyrPtrs pp = {};
    for (int l = 0; l < cfg->levels; l++) {
        if (grad_levels[l] && (((uintptr_t)grad_levels[l]) & 15) != 0) return LISO_EINVAL;
        pp.grad[l] = grad_levels[l];
    }
    const long total = (long)cfg->batch * cfg->h * cfg->w * (cfg->dim / 4);
    corr_pyramid_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(*cfg, pp, grad_fmap2);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

size_t liso_corr_bwd_features_workspace_bytes(const liso_corr_cfg* cfg) {
    if (!cfg_ok(cfg)) return 0;
    const Plan p = make_plan(cfg);
    const size_t g1 = p.splits1 > 1 ? (size_t)p.splits1 * cfg->batch * p.a.hw * cfg->dim * sizeof(float) : 0;
    const size_t g2 = p.splits2 > 1 ? (size_t)p.splits2 * cfg->batch * p.a.row0[cfg->levels] * cfg->dim * sizeof(float) : 0;
    const size_t need = g1 > g2 ? g1 : g2;  // (the two products run one after the other on the stream: one buffer)
    return need ? need : 16;
}

int liso_corr_bwd_features_f32(const liso_corr_cfg* cfg, int mode, const float* fmap1, const float* const* fmap2_levels,
                               const float* const* dvol_levels, float* grad_fmap1, float* const* grad_fmap2_levels, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!cfg_ok(cfg) || !fmap1 || !fmap2_levels || !dvol_levels || !grad_fmap1 || !grad_fmap2_levels) return LISO_EINVAL;
    if (mode != LISO_CONV_F32X3 && mode != LISO_CONV_F32) return LISO_EINVAL;
    Plan p = make_plan(cfg);
    if (!workspace || workspace_bytes < liso_corr_bwd_features_workspace_bytes(cfg)) return LISO_EWORKSPACE;
    BwdArgs& a = p.a;
    a.fmap1 = fmap1; a.g1 = grad_fmap1; a.part = (float*)workspace;
    bool vec = p.vec && (((uintptr_t)fmap1 | (uintptr_t)grad_fmap1 | (uintptr_t)workspace) & 15) == 0;
    for (int l = 0; l < cfg->levels; l++) {
        if (!fmap2_levels[l] || !dvol_levels[l] || !grad_fmap2_levels[l]) return LISO_EINVAL;
        if ((((uintptr_t)fmap2_levels[l] | (uintptr_t)grad_fmap2_levels[l]) & 15) != 0) return LISO_EINVAL;  // (rows of D floats: always whole vectors)
        if (((uintptr_t)dvol_levels[l] & 15) != 0) vec = false;
        a.f2[l] = fmap2_levels[l]; a.dvol[l] = dvol_levels[l]; a.g2[l] = grad_fmap2_levels[l];
    }
    if ((((uintptr_t)fmap1 | (uintptr_t)grad_fmap1 | (uintptr_t)workspace) & 15) != 0) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // G1: grad_fmap1
    a.splits = p.splits1; a.m_tiles = p.m_tiles1;
    rc = mode == LISO_CONV_F32X3 ? launch_gemm<LISO_CONV_F32X3, false>(a, vec, p.m_tiles1, st) : launch_gemm<LISO_CONV_F32, false>(a, vec, p.m_tiles1, st);
    if (rc != LISO_OK) return rc;
    if (a.splits > 1) {
        const long total = (long)a.batch * a.hw * (a.dim / 4);
        corr_bwd_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, 0);
    }
    // G2: grad_fmap2 of every level
    a.splits = p.splits2; a.m_tiles = p.m_tiles2;
    rc = mode == LISO_CONV_F32X3 ? launch_gemm<LISO_CONV_F32X3, true>(a, vec, p.m_tiles2, st) : launch_gemm<LISO_CONV_F32, true>(a, vec, p.m_tiles2, st);
    if (rc != LISO_OK) return rc;
    if (a.splits > 1) {
        const long total = (long)a.batch * a.row0[a.levels] * (a.dim / 4);
        corr_bwd_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, 1);
    }
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // extern "C"
