// Kabsch / weighted point-cloud alignment for gfx950 (MI355X).  C ABI + reference lines: include/liso_kabsch.h.
//
// Structure (HBM-bound: the cloud is read once, 20-24 B per point; everything else lives in registers/LDS):
//   kabsch_moments_kernel   per (point block, sample, slot chunk)
//       phase A  lanes = points : background weight  prod_s (1 - w_s(scale_bg))  -> bg + uniform-slot moments
//       phase B  lanes = slots  : every lane owns one box and walks the tile's points (LDS broadcast), keeping its 9
//                                 weighted moments in registers -- no cross-lane reduction per point, no [B,S,N,4]
//       fg weights (optional output) are staged per wave in LDS and written as coalesced 128-B row segments
//   kabsch_solve_kernel     per (sample, slot): fixed-order fp64 reduction of the block partials, epsilon rule,
//                           centred cross-covariance, 3x3 one-sided Jacobi SVD -> R = U V^T, T = [R | y - R x]
//   symm_ortho_{fwd,bwd}    the same Jacobi for liso.torch_symm_ortho (forward + the reference's analytic backward)
// Moments are accumulated relative to each slot's box centre, which keeps fp32 partial sums well conditioned
// (the centred covariance is shift invariant); all cross-block sums are fp64 in a fixed order (reproducible).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_kabsch.h"

namespace {

constexpr int kThreads = 256;
constexpr int kTile = 128;            // points per tile (32 per wave in phase B)
constexpr int kPtsPerWave = kTile / 4;
constexpr int kMaxBlocks = 512;       // point blocks per sample (round 5: 1024 -> 512; 256: the moments kernel itself slows down; one wave per slot of kabsch_solve_kernel walks the
                                      // blocks' partial sums: 938 rows of 36 B per slot took 38 us, more than the moments of 18 boxes)
constexpr int NM = LISO_KABSCH_NMOM;
constexpr float kPi = 3.14159265358979323846f;

struct BoxP {
    float cx, cy, cz, c, s, hl, hw, hh;  // centre, cos/sin(yaw), half extents (already scaled)
};

__device__ __forceinline__ float soft(float x, int softness) {
    // kabsch_mask.py:26-32: cauchy = 0.5 + atan(x)/pi ; or torch.sigmoid
    return softness == 0 ? 0.5f + (1.0f / kPi) * atanf(x) : 1.0f / (1.0f + expf(-x));
}

// kabsch_mask.py:161-228: point into the box frame (inverse of translate*rotate_z), per-axis soft inside test
__device__ __forceinline__ float box_weight(const BoxP& b, float px, float py, float pz, float slope, int softness) {
    const float dx = px - b.cx, dy = py - b.cy, dz = pz - b.cz;
    const float bx = b.c * dx + b.s * dy;
    const float by = -b.s * dx + b.c * dy;
    const float lx = slope * (b.hl - fabsf(bx));
    const float ly = slope * (b.hw - fabsf(by));
    const float lz = slope * (b.hh - fabsf(dz));
    return soft(lx, softness) * soft(ly, softness) * soft(lz, softness);
}

__device__ __forceinline__ BoxP load_box(const float* pos, const float* dims, const float* rot, int idx, float scale) {
    BoxP b;
    b.cx = pos[idx * 3 + 0]; b.cy = pos[idx * 3 + 1]; b.cz = pos[idx * 3 + 2];
    const float th = rot[idx];
    b.c = cosf(th); b.s = sinf(th);
    // kabsch_mask.py:293-295 dims * obj_dim_scale, then /2 in get_box_pixel_weights (:212-222)
    b.hl = dims[idx * 3 + 0] * scale / 2; b.hw = dims[idx * 3 + 1] * scale / 2; b.hh = dims[idx * 3 + 2] * scale / 2;
    return b;
}

__device__ __forceinline__ void accumulate(float* acc, float w, float x0, float x1, float y0, float y1) {
    acc[0] += w;
    acc[1] = fmaf(w, x0, acc[1]); acc[2] = fmaf(w, x1, acc[2]);
    acc[3] = fmaf(w, y0, acc[3]); acc[4] = fmaf(w, y1, acc[4]);
    const float wy0 = w * y0, wy1 = w * y1;
    acc[5] = fmaf(wy0, x0, acc[5]); acc[6] = fmaf(wy0, x1, acc[6]);
    acc[7] = fmaf(wy1, x0, acc[7]); acc[8] = fmaf(wy1, x1, acc[8]);
}

constexpr int kBgCache = 128;  // background-scale boxes of the first slots, kept in LDS (the rest are rebuilt from global memory)
struct MomLds {
    float px[kTile], py[kTile], pz[kTile], fx[kTile], fy[kTile], ok[kTile];
    BoxP bg[kBgCache];
    float wbuf[4][64][kPtsPerWave + 1];
    float red[4][NM][64];
    float red2[2][NM][kTile];
};

__global__ __launch_bounds__(kThreads) void kabsch_moments_kernel(liso_kabsch_cfg cfg, const float* __restrict__ points,
                                                                  const uint8_t* __restrict__ valid,
                                                                  const float* __restrict__ flow,
                                                                  const float* __restrict__ box_pos,
                                                                  const float* __restrict__ box_dims,
                                                                  const float* __restrict__ box_rot,
                                                                  float* __restrict__ fg_weights,
                                                                  float* __restrict__ partials, int tiles_per_block,
                                                                  const int* __restrict__ slot_count) {
    __shared__ MomLds L;
    const int b = blockIdx.y, chunk = blockIdx.z;
    const int N = cfg.n_points, S = cfg.n_slots;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // `slot_count` (fixed-slot callers): only the first cnt slots of the sample hold boxes; the others are parked where their mask is 0
    // in fp32 -- a factor of exactly 1 in the background product, moments below the epsilon rule's 1e-12 (kabsch_solve_kernel then
    // ignores them): skipping them changes no output bit, and the lanes are re-dealt: SV slots x (64 / SV) point sub-lanes per wave
    int cnt = S;
    if (slot_count) {
        cnt = slot_count[b];
        cnt = cnt < 0 ? 0 : (cnt > S ? S : cnt);
    }
    const int cnt_chunk = cnt - chunk * 64 < 0 ? 0 : (cnt - chunk * 64 > 64 ? 64 : cnt - chunk * 64);
    // (the deal follows the number of boxes: without slot_count that is n_slots -- a caller that passes exactly its boxes as slots gets
    // the deal a fixed-slot caller with the same count gets, and with it the same bits: the eager / captured box-mining paths of
    // liso_amd/trainer.py rely on that, tests/test_gpu_liso_loop.py)
    const int SV = cnt_chunk <= 16 ? 16 : (cnt_chunk <= 32 ? 32 : 64), NP = 64 / SV;
    const int slot_l = lane % SV, psub = lane / SV;
    const int slot = chunk * 64 + slot_l;
    const bool has_slot = slot < cnt;
    const float* bpos = box_pos + (size_t)b * S * 3;
    const float* bdim = box_dims + (size_t)b * S * 3;
    const float* brot = box_rot + (size_t)b * S;
    BoxP mybox = {};
    if (has_slot) mybox = load_box(bpos, bdim, brot, slot, cfg.scale_fg);
    if (chunk == 0)
        for (int q = tid; q < cnt && q < kBgCache; q += kThreads) L.bg[q] = load_box(bpos, bdim, brot, q, cfg.scale_bg);
    float acc[NM], accbg[NM], accuni[NM];
#pragma unroll
    for (int k = 0; k < NM; k++) acc[k] = accbg[k] = accuni[k] = 0.f;

    const int tile0 = blockIdx.x * tiles_per_block;
    for (int t = 0; t < tiles_per_block; t++) {
        const int base = (tile0 + t) * kTile;
        if (base >= N) break;
        if (tid < kTile) {
            const int n = base + tid;
            float px = 0.f, py = 0.f, pz = 0.f, fx = 0.f, fy = 0.f, ok = 0.f;
            if (n < N && valid[(size_t)b * N + n]) {
                const float* p = points + ((size_t)b * N + n) * cfg.point_stride;
                const float* f = flow + ((size_t)b * N + n) * cfg.flow_stride;
                px = p[0]; py = p[1]; pz = p[2]; fx = f[0]; fy = f[1]; ok = 1.f;
            }
            // padding / invalid rows: coordinates and flow mapped to zero (kabsch_mask.py:20-22, :410-416)
            L.px[tid] = px; L.py[tid] = py; L.pz[tid] = pz; L.fx[tid] = fx; L.fy[tid] = fy;
            L.ok[tid] = (n < N) ? (ok > 0.f ? 1.f : 0.5f) : 0.f;  // 1 valid, 0.5 padded-in-range, 0 beyond N
        }
        __syncthreads();
        // ---- phase A: background + uniform slots, lanes = points --------------------------------------------------
        if (chunk == 0 && tid < kTile) {
            const float okf = L.ok[tid];
            if (okf > 0.f) {
                const float px = L.px[tid], py = L.py[tid], pz = L.pz[tid];
                float prod = 1.f;
                for (int s = 0; s < cnt; s++) {
                    const BoxP bb = s < kBgCache ? L.bg[s] : load_box(bpos, bdim, brot, s, cfg.scale_bg);
                    prod *= 1.0f - box_weight(bb, px, py, pz, cfg.slope, cfg.softness);
                }
                // mask_fusing.py:4-6 then kabsch_mask.py:370-372: bg = 1 - (1 - prod)
                const float occ = 1.0f - prod;
                const float wbg = okf == 1.f ? 1.0f - occ : 0.f;  // invalid points: weight zeroed (:417-419)
                const float y0 = px + L.fx[tid], y1 = py + L.fy[tid];
                accumulate(accbg, wbg, px, py, y0, y1);
                accumulate(accuni, 1.0f, px, py, y0, y1);         // epsilon rule needs sums over ALL N rows
            }
        }
        // ---- phase B: foreground slots, lanes = slots ---------------------------------------------------------------
        {
            const int p0 = wave * kPtsPerWave;
            if (fg_weights && NP > 1)  // (lanes that hold no slot leave their rows of the staging buffer unwritten: zeros go out)
                for (int i = 0; i < kPtsPerWave; i++) L.wbuf[wave][lane][i] = 0.f;
            for (int ii = psub; ii < kPtsPerWave; ii += NP) {
                const int i = ii;
                const int p = p0 + i;
                const float okf = L.ok[p];
                float w = 0.f;
                if (has_slot && okf > 0.f) {
                    // padding rows sit at the origin (mapped there by the reference, :20-22): the RETURNED fg weight is
                    // the mask value of that point (:353-360), but it is zeroed before the alignment (:417-419)
                    const float px = L.px[p], py = L.py[p];
                    w = box_weight(mybox, px, py, L.pz[p], cfg.slope, cfg.softness);
                    if (okf == 1.f) {
                        const float x0 = px - mybox.cx, x1 = py - mybox.cy;
                        accumulate(acc, w, x0, x1, x0 + L.fx[p], x1 + L.fy[p]);
                    }
                }
                if (fg_weights && (has_slot || NP == 1)) L.wbuf[wave][slot_l][i] = w;
            }
            if (fg_weights) {
                __builtin_amdgcn_wave_barrier();
                // coalesced write-out: two slot rows per instruction, 32 consecutive points each
                const int half = lane >> 5, col = lane & 31;
                for (int r = 0; r < 64; r += 2) {
                    const int sl = chunk * 64 + r + half;
                    const int n = base + p0 + col;
                    if (sl < S && n < N) fg_weights[((size_t)b * S + sl) * N + n] = L.wbuf[wave][r + half][col];
                }
            }
        }
        __syncthreads();
    }
    // ---- block reduction (fixed order) -----------------------------------------------------------------------------
    // the point sub-lanes of a slot: fixed-order butterfly (lanes slot, slot + SV, ...); slots without a box: zeros
#pragma unroll
    for (int k = 0; k < NM; k++) {
        float v = acc[k];
        for (int o = SV; o < 64; o <<= 1) v += __shfl_xor(v, o);
        acc[k] = has_slot ? v : 0.f;
    }
    if (lane < SV) {
#pragma unroll
        for (int k = 0; k < NM; k++) L.red[wave][k][lane] = acc[k];
    }
    for (int l = SV + lane; l < 64 && lane < 64; l += 64) {
#pragma unroll
        for (int k = 0; k < NM; k++) L.red[wave][k][l] = 0.f;
    }
    if (tid < kTile) {
#pragma unroll
        for (int k = 0; k < NM; k++) { L.red2[0][k][tid] = accbg[k]; L.red2[1][k][tid] = accuni[k]; }
    }
    __syncthreads();
    const int nblk = gridDim.x;
    float* out = partials + ((size_t)b * nblk + blockIdx.x) * (S + 2) * NM;
    for (int idx = tid; idx < 64 * NM; idx += kThreads) {
        const int k = idx / 64, l = idx % 64;
        const int sl = chunk * 64 + l;
        if (sl < S) out[sl * NM + k] = (L.red[0][k][l] + L.red[1][k][l]) + (L.red[2][k][l] + L.red[3][k][l]);
    }
    if (chunk == 0 && tid < 2 * NM) {
        const int which = tid / NM, k = tid % NM;
        float s = 0.f;
        for (int i = 0; i < kTile; i++) s += L.red2[which][k][i];
        out[(S + which) * NM + k] = s;
    }
}

// ---- 3x3 one-sided (Hestenes) Jacobi SVD in fp64: A = U diag(d) V^T, d descending --------------------------------
struct Svd3 {
    double U[3][3], V[3][3], d[3];
};

__device__ void jacobi_svd3(const double A[3][3], Svd3& o) {
    double G[3][3], V[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { G[i][j] = A[i][j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 40; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; i++) { alpha += G[i][p] * G[i][p]; beta += G[i][q] * G[i][q]; gamma += G[i][p] * G[i][q]; }
                if (fabs(gamma) <= 1e-17 * sqrt(alpha * beta) || gamma == 0.0) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 3; i++) {
                    const double gp = G[i][p], gq = G[i][q];
                    G[i][p] = c * gp - s * gq; G[i][q] = s * gp + c * gq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double sg[3];
    for (int j = 0; j < 3; j++) sg[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++)
        for (int bq = 0; bq < 2 - a; bq++)
            if (sg[ord[bq]] < sg[ord[bq + 1]]) { const int t = ord[bq]; ord[bq] = ord[bq + 1]; ord[bq + 1] = t; }
    const double tiny = 1e-300 + 1e-15 * sg[ord[0]];
    int rank = 0;
    for (int j = 0; j < 3; j++) {
        const int c = ord[j];
        o.d[j] = sg[c];
        for (int i = 0; i < 3; i++) o.V[i][j] = V[i][c];
        if (sg[c] > tiny) {
            for (int i = 0; i < 3; i++) o.U[i][j] = G[i][c] / sg[c];
            rank = j + 1;
        }
    }
    // complete U for (numerically) zero singular values with a right-handed-if-V-is basis
    if (rank == 0) {
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.U[i][j] = o.V[i][j];
    } else if (rank == 1) {
        // any unit vector orthogonal to u0, then the cross product
        const double ax = fabs(o.U[0][0]), ay = fabs(o.U[1][0]), az = fabs(o.U[2][0]);
        double e[3] = {0, 0, 0};
        e[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.0;
        const double dot = e[0] * o.U[0][0] + e[1] * o.U[1][0] + e[2] * o.U[2][0];
        double v[3] = {e[0] - dot * o.U[0][0], e[1] - dot * o.U[1][0], e[2] - dot * o.U[2][0]};
        const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int i = 0; i < 3; i++) o.U[i][1] = v[i] / nv;
        rank = 2;
    }
    if (rank == 2) {
        const double detV = o.V[0][0] * (o.V[1][1] * o.V[2][2] - o.V[1][2] * o.V[2][1]) -
                            o.V[0][1] * (o.V[1][0] * o.V[2][2] - o.V[1][2] * o.V[2][0]) +
                            o.V[0][2] * (o.V[1][0] * o.V[2][1] - o.V[1][1] * o.V[2][0]);
        const double sgn = detV >= 0 ? 1.0 : -1.0;  // det(U) := det(V): no reflection through the null direction
        o.U[0][2] = sgn * (o.U[1][0] * o.U[2][1] - o.U[2][0] * o.U[1][1]);
        o.U[1][2] = sgn * (o.U[2][0] * o.U[0][1] - o.U[0][0] * o.U[2][1]);
        o.U[2][2] = sgn * (o.U[0][0] * o.U[1][1] - o.U[1][0] * o.U[0][1]);
    }
}

// per (sample, slot): reduce, epsilon rule, centre, solve, compose T.  One wave per slot, lanes split the blocks.
__global__ __launch_bounds__(64) void kabsch_solve_kernel(liso_kabsch_cfg cfg, const float* __restrict__ partials,
                                                          int nblk, const float* __restrict__ box_pos,
                                                          double* __restrict__ trafos, float* __restrict__ cum_wts) {
    const int S = cfg.n_slots;
    const int b = blockIdx.y, slot = blockIdx.x;  // slot in [0, S]  (S = background)
    const int lane = threadIdx.x;
    double m[NM], u[NM];
#pragma unroll
    for (int k = 0; k < NM; k++) m[k] = u[k] = 0.0;
    for (int blk = lane; blk < nblk; blk += 64) {
        const float* p = partials + ((size_t)b * nblk + blk) * (S + 2) * NM;
#pragma unroll
        for (int k = 0; k < NM; k++) { m[k] += (double)p[slot * NM + k]; u[k] += (double)p[(S + 1) * NM + k]; }
    }
#pragma unroll
    for (int k = 0; k < NM; k++)
        for (int o = 32; o > 0; o >>= 1) { m[k] += __shfl_xor(m[k], o); u[k] += __shfl_xor(u[k], o); }
    if (lane != 0) return;
    double cx = 0.0, cy = 0.0;  // the shift used while accumulating this slot
    if (slot < S) { cx = (double)box_pos[((size_t)b * S + slot) * 3 + 0]; cy = (double)box_pos[((size_t)b * S + slot) * 3 + 1]; }
    // cum_wts is an fp32 sum in the reference (kabsch_mask.py:453); the epsilon rule (:454-470) compares it to 1e-12
    double W = m[0];
    if ((float)W < 1e-12f) {
        // every one of the N rows (padding included, mapped to the origin) gets weight 1e-12
        const double eps = (double)1e-12f;
        for (int k = 0; k < NM; k++) m[k] = eps * u[k];
        // the uniform slot was accumulated unshifted; padded rows contribute zero vectors but count in N
        m[0] = eps * (double)cfg.n_points;
        W = m[0];
        cx = 0.0; cy = 0.0;
    }
    const double mx0 = m[1] / W, mx1 = m[2] / W, my0 = m[3] / W, my1 = m[4] / W;  // relative to the shift
    double A[3][3] = {{m[5] / W - my0 * mx0, m[6] / W - my0 * mx1, 0.0},
                      {m[7] / W - my1 * mx0, m[8] / W - my1 * mx1, 0.0},
                      {0.0, 0.0, 0.0}};  // z zeroed (:413,:417-420) -> rank <= 2
    Svd3 sv;
    jacobi_svd3(A, sv);
    double R[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i][j] = sv.U[i][0] * sv.V[j][0] + sv.U[i][1] * sv.V[j][1] + sv.U[i][2] * sv.V[j][2];
    const double X0 = mx0 + cx, X1 = mx1 + cy, Y0 = my0 + cx, Y1 = my1 + cy;
    double* T = trafos + ((size_t)b * (S + 1) + slot) * 16;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R[i][j];
        const double yi = i == 0 ? Y0 : (i == 1 ? Y1 : 0.0);
        T[i * 4 + 3] = yi - (R[i][0] * X0 + R[i][1] * X1);  // :493-495, z of both means is 0
    }
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
    cum_wts[(size_t)b * (S + 1) + slot] = (float)W;
}

__global__ void symm_ortho_fwd_kernel(const double* __restrict__ a, int n, double* __restrict__ r,
                                      double* __restrict__ u, double* __restrict__ vh, double* __restrict__ d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double A[3][3];
    for (int p = 0; p < 3; p++) for (int q = 0; q < 3; q++) A[p][q] = a[(size_t)i * 9 + p * 3 + q];
    Svd3 sv;
    jacobi_svd3(A, sv);
    for (int p = 0; p < 3; p++)
        for (int q = 0; q < 3; q++) {
            r[(size_t)i * 9 + p * 3 + q] = sv.U[p][0] * sv.V[q][0] + sv.U[p][1] * sv.V[q][1] + sv.U[p][2] * sv.V[q][2];
            u[(size_t)i * 9 + p * 3 + q] = sv.U[p][q];
            vh[(size_t)i * 9 + p * 3 + q] = sv.V[q][p];
        }
    for (int p = 0; p < 3; p++) d[(size_t)i * 3 + p] = sv.d[p];
}

// torch_symm_ortho/__init__.py:15-43 in closed form: grad_A = U (W - W^T) Vh, W = (U^T G V) / (d_k + d_l + delta_kl)
__global__ void symm_ortho_bwd_kernel(const double* __restrict__ g, const double* __restrict__ u,
                                      const double* __restrict__ vh, const double* __restrict__ d, int n,
                                      double* __restrict__ ga) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double G[3][3], U[3][3], Vh[3][3], D[3];
    for (int p = 0; p < 3; p++) {
        D[p] = d[(size_t)i * 3 + p];
        for (int q = 0; q < 3; q++) {
            G[p][q] = g[(size_t)i * 9 + p * 3 + q]; U[p][q] = u[(size_t)i * 9 + p * 3 + q]; Vh[p][q] = vh[(size_t)i * 9 + p * 3 + q];
        }
    }
    double M[3][3], Wm[3][3];
    for (int k = 0; k < 3; k++)
        for (int l = 0; l < 3; l++) {
            double s = 0.0;
            for (int p = 0; p < 3; p++)
                for (int q = 0; q < 3; q++) s += U[p][k] * G[p][q] * Vh[l][q];  // V[q][l] = Vh[l][q]
            M[k][l] = s / (D[k] + D[l] + (k == l ? 1.0 : 0.0));
        }
    for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) Wm[k][l] = M[k][l] - M[l][k];
    for (int p = 0; p < 3; p++)
        for (int q = 0; q < 3; q++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) s += U[p][k] * Wm[k][l] * Vh[l][q];
            ga[(size_t)i * 9 + p * 3 + q] = s;
        }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

inline bool cfg_ok(const liso_kabsch_cfg* c) {
    return c && c->batch >= 1 && c->n_points >= 0 && c->n_slots >= 0 && c->point_stride >= 3 && c->flow_stride >= 2 &&
           (c->softness == 0 || c->softness == 1);
}

inline int num_blocks(int n) {
    const int tiles = (n + kTile - 1) / kTile;
    return tiles < kMaxBlocks ? (tiles > 0 ? tiles : 1) : kMaxBlocks;
}

}  // namespace

extern "C" {

size_t liso_kabsch_workspace_bytes(const liso_kabsch_cfg* cfg) {
    if (!cfg_ok(cfg)) return 0;
    return (size_t)cfg->batch * num_blocks(cfg->n_points) * (cfg->n_slots + 2) * NM * sizeof(float);
}

int liso_kabsch_trafos_f32(const liso_kabsch_cfg* cfg, const float* points, const uint8_t* valid, const float* flow,
                           const float* box_pos, const float* box_dims, const float* box_rot, double* trafos,
                           float* cum_wts, float* fg_weights, void* workspace, size_t workspace_bytes, void* stream) {
    return liso_kabsch_trafos_counted_f32(cfg, points, valid, flow, box_pos, box_dims, box_rot, nullptr, trafos, cum_wts, fg_weights,
                                          workspace, workspace_bytes, stream);
}

int liso_kabsch_trafos_counted_f32(const liso_kabsch_cfg* cfg, const float* points, const uint8_t* valid, const float* flow,
                                   const float* box_pos, const float* box_dims, const float* box_rot, const int32_t* slot_count,
                                   double* trafos, float* cum_wts, float* fg_weights, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!cfg_ok(cfg) || !trafos || !cum_wts || !workspace) return LISO_EINVAL;
    if (cfg->n_points > 0 && (!points || !valid || !flow)) return LISO_EINVAL;
    if (cfg->n_slots > 0 && (!box_pos || !box_dims || !box_rot)) return LISO_EINVAL;
    if (workspace_bytes < liso_kabsch_workspace_bytes(cfg)) return LISO_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = num_blocks(cfg->n_points);
    const int tiles = (cfg->n_points + kTile - 1) / kTile;
    const int tpb = (tiles + nblk - 1) / nblk > 0 ? (tiles + nblk - 1) / nblk : 1;
    const int chunks = (cfg->n_slots + 63) / 64 > 0 ? (cfg->n_slots + 63) / 64 : 1;
    kabsch_moments_kernel<<<dim3(nblk, cfg->batch, chunks), kThreads, 0, st>>>(*cfg, points, valid, flow, box_pos,
                                                                               box_dims, box_rot, fg_weights,
                                                                               (float*)workspace, tpb, slot_count);
    kabsch_solve_kernel<<<dim3(cfg->n_slots + 1, cfg->batch), 64, 0, st>>>(*cfg, (const float*)workspace, nblk, box_pos,
                                                                           trafos, cum_wts);
    return check_launch();
}

int liso_symm_ortho_fwd_f64(const double* a, int n, double* r, double* u, double* vh, double* d, void* stream) {
    if (n < 0) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    if (!a || !r || !u || !vh || !d) return LISO_EINVAL;
    symm_ortho_fwd_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(a, n, r, u, vh, d);
    return check_launch();
}

int liso_symm_ortho_bwd_f64(const double* grad_r, const double* u, const double* vh, const double* d, int n,
                            double* grad_a, void* stream) {
    if (n < 0) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    if (!grad_r || !u || !vh || !d || !grad_a) return LISO_EINVAL;
    symm_ortho_bwd_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(grad_r, u, vh, d, n, grad_a);
    return check_launch();
}

}  // extern "C"
