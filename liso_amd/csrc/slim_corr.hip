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

template <int VEC>  // VEC float4 per lane: D = 128 * VEC
__global__ __launch_bounds__(64 * kWavesPerBlock) void corr_lookup_fwd_kernel(liso_corr_cfg c,
                                                                              const float* __restrict__ fmap1,
                                                                              LevelPtrs lp, const float* __restrict__ coords,
                                                                              float* __restrict__ out) {
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
        float4* wrow = reinterpret_cast<float4*>(&T[wave][lane][0]);
#pragma unroll
        for (int k = 0; k < 8; k++) wrow[k] = make_float4(acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]);
    }
    __builtin_amdgcn_wave_barrier();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 32; j++) tot += T[wave][half * 32 + j][g];  // fixed order: bit reproducible
    P[wave][2 * g + half] = tot;
    __builtin_amdgcn_wave_barrier();
    const int W7 = 2 * c.radius + 1;
    const int a = lane / W7, bb = lane % W7;  // a: x offset index, bb: y offset index (corr.py:31-35 order)
    if (lane < W7 * W7) {
        const float inv = 1.0f / sqrtf((float)D);  // corr.py:56
        const float p00 = P[wave][bb * 8 + a], p10 = P[wave][bb * 8 + a + 1];
        const float p01 = P[wave][(bb + 1) * 8 + a], p11 = P[wave][(bb + 1) * 8 + a + 1];
        const float v = (1.f - pt.fx) * (1.f - pt.fy) * p00 + pt.fx * (1.f - pt.fy) * p10 + (1.f - pt.fx) * pt.fy * p01 +
                        pt.fx * pt.fy * p11;
        const int C = c.levels * W7 * W7;
        out[((size_t)b * hw + pix) * C + lvl * W7 * W7 + lane] = v * inv;
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
