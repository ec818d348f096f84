// Per-box steps of the box mining between the clustering kernels and the detector targets (C ABI + the reference lines each entry
// replaces: include/liso_box_mining.h).  Everything here is tiny (K <= a few hundred boxes per sample): the point is not bandwidth
// but that ONE launch replaces 10-40 framework launches, that nothing calls a scan / sort library (their memset nodes do not
// survive in a replayed hipGraph on this runtime) and that no step needs a host read -- so the whole mining stage of a sweep pair
// can be captured into a hipGraph (liso_amd/trainer.py, stage B).
//
//   scan_*                 inclusive int32 row scan: block sums -> one-block scan of the sums -> local scan + offset (fixed order)
//   boxes_from_regions     one thread per (sample, region)
//   filter_compact         one wavefront per sample: filters, ballot-prefix stable compaction, every destination slot written once
//   box_motion             one thread per box: affine 4x4 algebra in fp64 (closed-form inverses)
//   nms_prepare / finish   one wavefront per sample: rank-by-counting stable order (K^2 / 64 compares per lane), in-place permutation
//                          through a scratch copy; post-NMS selection with LDS flags + ballot prefix
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_box_mining.h"
#include "../../include/liso_iou3d.h"

namespace {

constexpr int kScanThreads = 256;
constexpr int kScanPer = 8;                            // elements per thread
constexpr int kScanChunk = kScanThreads * kScanPer;    // elements per block
constexpr int32_t kUnknownClass = 2147483647;          // shape_utils.py:15
constexpr int32_t kInvalidClass = kUnknownClass - 1;   // shape_utils.py:16

int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

// ---- scan -----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
    // exclusive scan of one int per thread over a 256-thread block (4 waves): DPP-free shuffles + 4 LDS words
    __shared__ int wsum[kScanThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; w++) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(kScanThreads) void scan_block_sums_kernel(const int32_t* __restrict__ in, long n, int nblk,
                                                                       int32_t* __restrict__ sums) {
    const long row = blockIdx.y, base = (long)blockIdx.x * kScanChunk + (long)threadIdx.x * kScanPer;
    const int32_t* p = in + row * n;
    int s = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; e++)
        if (base + e < n) s += p[base + e];
    int tot;
    block_exclusive_scan(s, &tot);
    if (threadIdx.x == 0) sums[row * nblk + blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanThreads) void scan_sums_kernel(int32_t* __restrict__ sums, int nblk) {
    // exclusive scan of one row of block sums, in place; chunks of 256 with a running carry
    int32_t* p = sums + (long)blockIdx.x * nblk;
    int carry = 0;
    for (int c0 = 0; c0 < nblk; c0 += kScanThreads) {
        const int i = c0 + threadIdx.x;
        const int v = i < nblk ? p[i] : 0;
        int tot;
        const int ex = block_exclusive_scan(v, &tot);
        if (i < nblk) p[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const int32_t* __restrict__ in, long n, int nblk,
                                                                  const int32_t* __restrict__ offs, int32_t* __restrict__ out) {
    const long row = blockIdx.y, base = (long)blockIdx.x * kScanChunk + (long)threadIdx.x * kScanPer;
    const int32_t* p = in + row * n;
    int v[kScanPer], s = 0;
#pragma unroll
    for (int e = 0; e < kScanPer; e++) {
        v[e] = base + e < n ? p[base + e] : 0;
        s += v[e];
    }
    int tot;
    int run = block_exclusive_scan(s, &tot) + offs[row * nblk + blockIdx.x];
    int32_t* q = out + row * n;
#pragma unroll
    for (int e = 0; e < kScanPer; e++) {
        run += v[e];
        if (base + e < n) q[base + e] = run;
    }
}

// ---- boxes from regions -----------------------------------------------------------------------------------------------------------
__global__ void boxes_from_regions_kernel(const double* __restrict__ props, long total, const float* __restrict__ row_coords, int gx,
                                          const float* __restrict__ col_coords, int gy, double ppm_x, double ppm_y,
                                          float* __restrict__ center, double* __restrict__ dims, double* __restrict__ rot,
                                          float* __restrict__ dims_f32, float* __restrict__ rot_f32) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double* p = props + i * 5;
    const long hi = (gx < gy ? gx : gy) - 1;
    // centroid -> int64 (truncation toward zero), clipped, then the centre of THAT pillar (flow_cluster_detector.py:176-180)
    long r = (long)p[0], c = (long)p[1];
    r = r < 0 ? 0 : (r > hi ? hi : r);
    c = c < 0 ? 0 : (c > hi ? hi : c);
    center[i * 2 + 0] = row_coords[r];
    center[i * 2 + 1] = col_coords[c];
    const double d0 = p[3] * 1.0 / ppm_x, d1 = p[4] * 1.0 / ppm_y;
    dims[i * 2 + 0] = d0;
    dims[i * 2 + 1] = d1;
    rot[i] = p[2];
    dims_f32[i * 2 + 0] = (float)d0;
    dims_f32[i * 2 + 1] = (float)d1;
    rot_f32[i] = (float)p[2];
}

// ---- filters + compaction ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void filter_compact_kernel(const liso_mine_filter_cfg c, const int64_t* __restrict__ num_labels,
                                                            const float* __restrict__ center, const double* __restrict__ dims2,
                                                            const double* __restrict__ rot_in, const int64_t* __restrict__ num_pts,
                                                            const float* __restrict__ fit_z, const float* __restrict__ fit_h,
                                                            float* __restrict__ pos, double* __restrict__ dims, double* __restrict__ rot,
                                                            double* __restrict__ probs, double* __restrict__ velo,
                                                            uint8_t* __restrict__ valid, int32_t* __restrict__ class_id,
                                                            int32_t* __restrict__ difficulty, int32_t* __restrict__ counts,
                                                            float* __restrict__ kpos, float* __restrict__ kdims, float* __restrict__ krot) {
    const int b = blockIdx.x, lane = threadIdx.x, K = c.k;
    const long o = (long)b * K;
    const long nl = num_labels[b];
    int run = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int k = k0 + lane;
        bool ok = false;
        double d0 = 0.0, d1 = 0.0, hz = 0.0, th = 0.0;
        float cx = 0.f, cy = 0.f, fz = 0.f;
        if (k < K) {
            d0 = dims2[(o + k) * 2];
            d1 = dims2[(o + k) * 2 + 1];
            hz = (double)fit_h[o + k];
            th = rot_in[o + k];
            cx = center[(o + k) * 2];
            cy = center[(o + k) * 2 + 1];
            fz = fit_z[o + k];
            const bool exists = k < nl;
            const bool enough = num_pts[o + k] >= (int64_t)c.min_points;
            const double aspect = d0 / fmax(d1, 0.001);
            const bool aspect_ok = aspect <= c.aspect_ratio_max;
            const bool not_large = d0 <= c.max_box_len_m;
            const bool foot_ok = d0 * d1 > c.min_box_area_m2;
            const bool vol_ok = (d0 * d1) * hz > c.min_box_volume_m3;
            ok = exists && enough && aspect_ok && not_large && foot_ok && vol_ok;
        }
        const unsigned long long m = __ballot(ok);
        const int dst = run + __popcll(m & ((1ull << lane) - 1ull));
        if (ok) {
            const long q = o + dst;
            pos[q * 3] = cx, pos[q * 3 + 1] = cy, pos[q * 3 + 2] = fz;
            dims[q * 3] = d0, dims[q * 3 + 1] = d1, dims[q * 3 + 2] = hz;
            rot[q] = th;
            probs[q] = 1.0;
            velo[q] = 0.0;
            valid[q] = 1;
            class_id[q] = kUnknownClass;
            difficulty[q] = 1;
            kpos[q * 3] = cx, kpos[q * 3 + 1] = cy, kpos[q * 3 + 2] = fz;
            kdims[q * 3] = (float)d0, kdims[q * 3 + 1] = (float)d1, kdims[q * 3 + 2] = (float)hz;
            krot[q] = (float)th;
        }
        run += __popcll(m);
    }
    for (int k = run + lane; k < K; k += 64) {  // the unused slots: padding values, each written once
        const long q = o + k;
        pos[q * 3] = 0.f, pos[q * 3 + 1] = 0.f, pos[q * 3 + 2] = 0.f;
        dims[q * 3] = 0.0, dims[q * 3 + 1] = 0.0, dims[q * 3 + 2] = 0.0;
        rot[q] = 0.0, probs[q] = 0.0, velo[q] = 0.0;
        valid[q] = 0;
        class_id[q] = kInvalidClass;
        difficulty[q] = kInvalidClass;
        const float park = c.park_invalid ? 1e6f : 0.f;
        kpos[q * 3] = park, kpos[q * 3 + 1] = park, kpos[q * 3 + 2] = park;
        kdims[q * 3] = 0.f, kdims[q * 3 + 1] = 0.f, kdims[q * 3 + 2] = 0.f;
        krot[q] = 0.f;
    }
    if (lane == 0) counts[b] = run;
}

// ---- box motion -------------------------------------------------------------------------------------------------------------------
struct Aff {  // affine transform: rotation / linear part a[3][3], translation t[3]
    double a[3][3], t[3];
};

__device__ __forceinline__ Aff aff_load(const double* __restrict__ m) {  // row-major 4x4, last row (0, 0, 0, 1)
    Aff r;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = m[i * 4 + j];
        r.t[i] = m[i * 4 + 3];
    }
    return r;
}

__device__ __forceinline__ Aff aff_mul(const Aff& x, const Aff& y) {  // x * y
    Aff r;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j];
        r.t[i] = x.a[i][0] * y.t[0] + x.a[i][1] * y.t[1] + x.a[i][2] * y.t[2] + x.t[i];
    }
    return r;
}

__device__ __forceinline__ Aff aff_inv(const Aff& x) {  // cofactor inverse of the 3x3 part, then -A^-1 t
    Aff r;
    const double (*a)[3] = x.a;
    const double c00 = a[1][1] * a[2][2] - a[1][2] * a[2][1], c01 = a[1][2] * a[2][0] - a[1][0] * a[2][2],
                 c02 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const double det = a[0][0] * c00 + a[0][1] * c01 + a[0][2] * c02;
    const double id = 1.0 / det;
    r.a[0][0] = c00 * id;
    r.a[1][0] = c01 * id;
    r.a[2][0] = c02 * id;
    r.a[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id;
    r.a[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id;
    r.a[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id;
    r.a[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id;
    r.a[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id;
    r.a[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id;
#pragma unroll
    for (int i = 0; i < 3; i++) r.t[i] = -(r.a[i][0] * x.t[0] + r.a[i][1] * x.t[1] + r.a[i][2] * x.t[2]);
    return r;
}

__global__ void box_motion_kernel(const double* __restrict__ trafos, int batch, int s, const float* __restrict__ pos,
                                  double* __restrict__ rot, double* __restrict__ velo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * s) return;
    const int b = i / s, k = i - b * s;
    const double* base = trafos + (long)b * (s + 1) * 16;
    const Aff fg = aff_load(base + (long)k * 16), bg = aff_load(base + (long)s * 16);
    const double th = rot[i];
    Aff box;  // torch_compose_matrix: translation * rotation about z (torch_transformation.py:16-62)
    const double cs = cos(th), sn = sin(th);
    box.a[0][0] = cs, box.a[0][1] = -sn, box.a[0][2] = 0.0;
    box.a[1][0] = sn, box.a[1][1] = cs, box.a[1][2] = 0.0;
    box.a[2][0] = 0.0, box.a[2][1] = 0.0, box.a[2][2] = 1.0;
    box.t[0] = (double)pos[(long)i * 3], box.t[1] = (double)pos[(long)i * 3 + 1], box.t[2] = (double)pos[(long)i * 3 + 2];
    // b0_dT_b1 = inv(T_box) inv(T_bg) (T_fg T_box)   (shape_utils.py:583-605)
    const Aff m = aff_mul(aff_mul(aff_inv(box), aff_inv(bg)), aff_mul(fg, box));
    rot[i] = th + atan2(m.t[1], m.t[0]);
    velo[i] = sqrt(m.t[0] * m.t[0] + m.t[1] * m.t[1] + m.t[2] * m.t[2]);
}

// ---- NMS preparation / selection ------------------------------------------------------------------------------------------------------
struct Slot {  // scratch copy of one box slot (72 B)
    double dims[3], rot, probs, velo;
    float pos[3];
    int32_t class_id, difficulty;
    uint8_t valid, pad[3];
};

__global__ __launch_bounds__(64) void nms_prepare_kernel(int k, int pre_nms_max, float* __restrict__ pos, double* __restrict__ dims,
                                                         double* __restrict__ rot, double* __restrict__ probs, double* __restrict__ velo,
                                                         uint8_t* __restrict__ valid, int32_t* __restrict__ class_id,
                                                         int32_t* __restrict__ difficulty, float* __restrict__ dense,
                                                         uint8_t* __restrict__ enters, Slot* __restrict__ scratch) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const long o = (long)b * k;
    Slot* sc = scratch + o;
    for (int i = lane; i < k; i += 64) {
        Slot s;
        const long q = o + i;
        s.pos[0] = pos[q * 3], s.pos[1] = pos[q * 3 + 1], s.pos[2] = pos[q * 3 + 2];
        s.dims[0] = dims[q * 3], s.dims[1] = dims[q * 3 + 1], s.dims[2] = dims[q * 3 + 2];
        s.rot = rot[q], s.probs = probs[q], s.velo = velo[q];
        s.class_id = class_id[q], s.difficulty = difficulty[q], s.valid = valid[q];
        s.pad[0] = s.pad[1] = s.pad[2] = 0;
        sc[i] = s;
    }
    __syncthreads();  // (one wavefront: orders the scratch stores before the reads of other lanes)
    for (int i = lane; i < k; i += 64) {
        // position in the stable descending order of key = valid ? probs : -inf.  A NaN confidence (a diverging detector) compares
        // false both ways and would give several slots one rank -- slots overwritten, others stale: NaN ranks as -inf (behind every
        // number, stable among themselves), so the order stays a permutation like the reference's torch.argsort
        const bool vi = sc[i].valid != 0;
        float ki = vi ? (float)sc[i].probs : -INFINITY;
        ki = ki != ki ? -INFINITY : ki;
        int rank = 0;
        for (int j = 0; j < k; j++) {
            float kj = sc[j].valid ? (float)sc[j].probs : -INFINITY;
            kj = kj != kj ? -INFINITY : kj;
            rank += (kj > ki || (kj == ki && j < i)) ? 1 : 0;
        }
        const Slot s = sc[i];
        const long q = o + rank;
        pos[q * 3] = s.pos[0], pos[q * 3 + 1] = s.pos[1], pos[q * 3 + 2] = s.pos[2];
        dims[q * 3] = s.dims[0], dims[q * 3 + 1] = s.dims[1], dims[q * 3 + 2] = s.dims[2];
        rot[q] = s.rot, probs[q] = s.probs, velo[q] = s.velo;
        class_id[q] = s.class_id, difficulty[q] = s.difficulty, valid[q] = s.valid;
        const bool in = vi && (pre_nms_max <= 0 || rank < pre_nms_max);
        enters[q] = in ? 1 : 0;
        float* dn = dense + q * 7;
        if (in) {
            dn[0] = s.pos[0], dn[1] = s.pos[1], dn[2] = s.pos[2];
            dn[3] = (float)s.dims[0], dn[4] = (float)s.dims[1], dn[5] = (float)s.dims[2];
            dn[6] = (float)s.rot;
        } else {  // far away, tiny, disjoint: overlaps nothing (nms_iou.py perform_nms_on_shapes_padded)
            dn[0] = 1e6f + 10.0f * (float)rank, dn[1] = 1e6f, dn[2] = 0.f;
            dn[3] = 1e-3f, dn[4] = 1e-3f, dn[5] = 1e-3f, dn[6] = 0.f;
        }
    }
}

constexpr int kMaxNmsSlots = 16384;

__global__ __launch_bounds__(64) void nms_finish_kernel(int b, int k, int max_boxes, const int64_t* __restrict__ keep,
                                                        const int32_t* __restrict__ num, const uint8_t* __restrict__ enters,
                                                        float* __restrict__ pos, double* __restrict__ dims, double* __restrict__ rot,
                                                        double* __restrict__ probs, double* __restrict__ velo, uint8_t* __restrict__ valid,
                                                        int32_t* __restrict__ class_id, int32_t* __restrict__ difficulty,
                                                        float* __restrict__ t_pos, float* __restrict__ t_dims, float* __restrict__ t_rot,
                                                        uint8_t* __restrict__ t_valid) {
    __shared__ uint8_t hit[kMaxNmsSlots];
    const int lane = threadIdx.x;
    const long o = (long)b * k;
    for (int i = lane; i < k; i += 64) hit[i] = 0;
    __syncthreads();
    const int n = min(max(num[0], 0), k);
    for (int i = lane; i < n; i += 64) {
        const long j = keep[i];
        if (j >= 0 && j < k) hit[j] = 1;
    }
    __syncthreads();
    int run = 0;
    for (int k0 = 0; k0 < k; k0 += 64) {
        const int i = k0 + lane;
        const bool kept = i < k && hit[i] && enters[o + i] && valid[o + i];
        const unsigned long long m = __ballot(kept);
        const int rank = run + __popcll(m & ((1ull << lane) - 1ull)) + 1;  // 1-based, like cumsum
        run += __popcll(m);
        if (i >= k) continue;
        const bool ok = kept && rank <= max_boxes;
        const long q = o + i;
        if (!ok) {  // Shape.set_padding_val_to(0.0) (shape_utils.py:439-462)
            pos[q * 3] = 0.f, pos[q * 3 + 1] = 0.f, pos[q * 3 + 2] = 0.f;
            dims[q * 3] = 0.0, dims[q * 3 + 1] = 0.0, dims[q * 3 + 2] = 0.0;
            rot[q] = 0.0, probs[q] = 0.0, velo[q] = 0.0;
            class_id[q] = kInvalidClass, difficulty[q] = kInvalidClass;
        }
        valid[q] = ok ? 1 : 0;
        t_valid[q] = ok ? 1 : 0;
        t_pos[q * 3] = pos[q * 3], t_pos[q * 3 + 1] = pos[q * 3 + 1], t_pos[q * 3 + 2] = pos[q * 3 + 2];
#pragma unroll
        for (int e = 0; e < 3; e++) t_dims[q * 3 + e] = fmaxf((float)dims[q * 3 + e], 1e-3f);
        t_rot[q] = (float)rot[q];
    }
}

}  // namespace

extern "C" {

size_t liso_scan_workspace_bytes(int batch, long n) {
    if (batch <= 0 || n <= 0) return 0;
    const long nblk = (n + kScanChunk - 1) / kScanChunk;
    return (size_t)batch * nblk * sizeof(int32_t);
}

int liso_scan_inclusive_i32(const int32_t* in, int batch, long n, int32_t* out, void* workspace, size_t workspace_bytes,
                            void* stream) {
    if (!in || !out || !workspace || batch <= 0 || n <= 0) return LISO_EINVAL;
    if (workspace_bytes < liso_scan_workspace_bytes(batch, n)) return LISO_EWORKSPACE;
    const long nblk = (n + kScanChunk - 1) / kScanChunk;
    if (nblk > 65535l * 64) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int32_t* sums = (int32_t*)workspace;
    scan_block_sums_kernel<<<dim3((unsigned)nblk, batch), kScanThreads, 0, st>>>(in, n, (int)nblk, sums);
    scan_sums_kernel<<<batch, kScanThreads, 0, st>>>(sums, (int)nblk);
    scan_apply_kernel<<<dim3((unsigned)nblk, batch), kScanThreads, 0, st>>>(in, n, (int)nblk, sums, out);
    return check_launch();
}

int liso_mine_boxes_from_regions(const double* props, int batch, int k, const float* row_coords, int gx, const float* col_coords,
                                 int gy, float ppm_x, float ppm_y, float* center, double* dims, double* rot, float* dims_f32,
                                 float* rot_f32, void* stream) {
    if (!props || !row_coords || !col_coords || !center || !dims || !rot || !dims_f32 || !rot_f32) return LISO_EINVAL;
    if (batch <= 0 || k <= 0 || gx <= 0 || gy <= 0) return LISO_EINVAL;
    const long total = (long)batch * k;
    boxes_from_regions_kernel<<<(unsigned)((total + 127) / 128), 128, 0, (hipStream_t)stream>>>(
        props, total, row_coords, gx, col_coords, gy, (double)ppm_x, (double)ppm_y, center, dims, rot, dims_f32, rot_f32);
    return check_launch();
}

int liso_mine_filter_compact(const liso_mine_filter_cfg* cfg, const int64_t* num_labels, const float* center, const double* dims2,
                             const double* rot_in, const int64_t* num_pts, const float* fit_z, const float* fit_h, float* pos,
                             double* dims, double* rot, double* probs, double* velo, uint8_t* valid, int32_t* class_id,
                             int32_t* difficulty, int32_t* counts, float* kabsch_pos, float* kabsch_dims, float* kabsch_rot,
                             void* stream) {
    if (!cfg || cfg->batch <= 0 || cfg->k <= 0) return LISO_EINVAL;
    if (!num_labels || !center || !dims2 || !rot_in || !num_pts || !fit_z || !fit_h || !pos || !dims || !rot || !probs || !velo ||
        !valid || !class_id || !difficulty || !counts || !kabsch_pos || !kabsch_dims || !kabsch_rot)
        return LISO_EINVAL;
    filter_compact_kernel<<<cfg->batch, 64, 0, (hipStream_t)stream>>>(*cfg, num_labels, center, dims2, rot_in, num_pts, fit_z, fit_h, pos,
                                                                     dims, rot, probs, velo, valid, class_id, difficulty, counts,
                                                                     kabsch_pos, kabsch_dims, kabsch_rot);
    return check_launch();
}

int liso_mine_box_motion(const double* trafos, int batch, int s, const float* pos, double* rot, double* velo, void* stream) {
    if (!trafos || !pos || !rot || !velo || batch <= 0 || s <= 0) return LISO_EINVAL;
    const int total = batch * s;
    box_motion_kernel<<<(total + 63) / 64, 64, 0, (hipStream_t)stream>>>(trafos, batch, s, pos, rot, velo);
    return check_launch();
}

size_t liso_mine_nms_workspace_bytes(int batch, int k) {
    if (batch <= 0 || k <= 0) return 0;
    return (size_t)batch * k * sizeof(Slot);
}

int liso_mine_nms_prepare(int batch, int k, int pre_nms_max, float* pos, double* dims, double* rot, double* probs, double* velo,
                          uint8_t* valid, int32_t* class_id, int32_t* difficulty, float* dense, uint8_t* enters, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (batch <= 0 || k <= 0 || !pos || !dims || !rot || !probs || !velo || !valid || !class_id || !difficulty || !dense || !enters ||
        !workspace)
        return LISO_EINVAL;
    if (workspace_bytes < liso_mine_nms_workspace_bytes(batch, k)) return LISO_EWORKSPACE;
    if (((uintptr_t)workspace & 7) != 0) return LISO_EINVAL;
    nms_prepare_kernel<<<batch, 64, 0, (hipStream_t)stream>>>(k, pre_nms_max, pos, dims, rot, probs, velo, valid, class_id, difficulty,
                                                             dense, enters, (Slot*)workspace);
    return check_launch();
}

int liso_mine_nms_finish(int b, int k, int max_boxes, const int64_t* keep, const int32_t* num, const uint8_t* enters, float* pos,
                         double* dims, double* rot, double* probs, double* velo, uint8_t* valid, int32_t* class_id,
                         int32_t* difficulty, float* t_pos, float* t_dims, float* t_rot, uint8_t* t_valid, void* stream) {
    if (b < 0 || k <= 0 || k > kMaxNmsSlots || !keep || !num || !enters || !pos || !dims || !rot || !probs || !velo || !valid ||
        !class_id || !difficulty || !t_pos || !t_dims || !t_rot || !t_valid)
        return LISO_EINVAL;
    nms_finish_kernel<<<1, 64, 0, (hipStream_t)stream>>>(b, k, max_boxes, keep, num, enters, pos, dims, rot, probs, velo, valid, class_id,
                                                        difficulty, t_pos, t_dims, t_rot, t_valid);
    return check_launch();
}

}  // extern "C"
