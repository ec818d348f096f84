// Per-point SLIM training decoder and its point-wise loss terms for gfx950.  C ABI + reference lines: include/liso_slim_decode.h.
//
// Everything here is one thread per point row (HBM-bound, 50-150 B per row); the reductions (masked means) are block partials in
// fp64 merged in a fixed order by the last launch of the call -- no atomics, bit reproducible.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/liso_iou3d.h"
#include "../../include/liso_slim_decode.h"

namespace {

constexpr int kThreads = 256;
constexpr int kRedBlocks = 512;  // block partials of the masked means

inline int check_launch() { return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH; }

// ---- the decode of one point (head_decoder.py:517-717 on one row) --------------------------------------------------------------
struct Consts {
    float c[4];  // value of a logit channel that is forced ON / OFF
};

// artificial_logit_network_output (:779-955), sequentially as the reference: a forced channel becomes a constant map and enters
// the extrema of the channels decided after it as that constant
__device__ __forceinline__ Consts logit_consts(const liso_slim_decode_cfg& c, const float* __restrict__ ext) {
    Consts k;
    float mx[4], mn[4];
    const bool need = c.logit_mode[1] != LISO_DECODE_NET || c.logit_mode[2] != LISO_DECODE_NET || c.logit_mode[3] != LISO_DECODE_NET;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        mx[i] = need ? ext[i] : 0.f;
        mn[i] = need ? ext[4 + i] : 0.f;
        k.c[i] = 0.f;
    }
    if (c.logit_mode[0] == LISO_DECODE_ON) k.c[0] = 0.f;
    if (c.logit_mode[0] == LISO_DECODE_OFF) k.c[0] = -100.f;
    if (c.logit_mode[1] != LISO_DECODE_NET) {  // static: max(dynamic, ground) +- 100
        k.c[1] = fmaxf(mx[2], mx[3]) + (c.logit_mode[1] == LISO_DECODE_ON ? 100.f : -100.f);
        mx[1] = mn[1] = k.c[1];
    }
    if (c.logit_mode[2] != LISO_DECODE_NET) {  // dynamic: ON max(static, ground) + 100; OFF min(static, ground) - 100
        k.c[2] = c.logit_mode[2] == LISO_DECODE_ON ? fmaxf(mx[1], mx[3]) + 100.f : fminf(mn[1], mn[3]) - 100.f;
        mx[2] = mn[2] = k.c[2];
    }
    if (c.logit_mode[3] != LISO_DECODE_NET)
        k.c[3] = c.logit_mode[3] == LISO_DECODE_ON ? fmaxf(mx[1], mx[2]) + 100.f : fminf(mn[1], mn[2]) - 100.f;
    return k;
}

struct Point {
    float logit[4];    // disappearing, static, dynamic, ground after modes and defaults
    float p[3];        // softmax(static, dynamic, ground)
    float sf[2], df[2];
    bool live_logit[4];  // gradient reaches the raw channel
    bool live_sf, live_df;
    bool filled;
};

__device__ __forceinline__ Point decode(const liso_slim_decode_cfg& c, const Consts& k, const float* __restrict__ r, bool filled) {
    Point q;
    q.filled = filled;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool net = c.logit_mode[i] == LISO_DECODE_NET;
        q.logit[i] = net ? r[i] : k.c[i];
        q.live_logit[i] = net;
    }
    q.sf[0] = c.static_flow_zero ? 0.f : r[4];
    q.sf[1] = c.static_flow_zero ? 0.f : r[5];
    q.df[0] = c.dynamic_flow_zero ? 0.f : r[6];
    q.df[1] = c.dynamic_flow_zero ? 0.f : r[7];
    q.live_sf = !c.static_flow_zero;
    q.live_df = !c.dynamic_flow_zero;
    if (!filled) {  // defaults at unfilled pillars (:566-590)
        if (c.overwrite_logits) {
            q.logit[0] = -100.f;
            q.logit[1] = c.logit_mode[1] == LISO_DECODE_OFF ? -100.f : 0.f;
            q.logit[2] = c.logit_mode[2] == LISO_DECODE_ON ? 0.f : -100.f;
            q.logit[3] = c.logit_mode[3] == LISO_DECODE_ON ? 0.f : -100.f;
#pragma unroll
            for (int i = 0; i < 4; i++) q.live_logit[i] = false;
        }
        if (c.overwrite_flow) {
            q.sf[0] = q.sf[1] = q.df[0] = q.df[1] = 0.f;
            q.live_sf = q.live_df = false;
        }
    }
    const float m = fmaxf(q.logit[1], fmaxf(q.logit[2], q.logit[3]));
    const float e0 = expf(q.logit[1] - m), e1 = expf(q.logit[2] - m), e2 = expf(q.logit[3] - m);
    const float s = e0 + e1 + e2;
    q.p[0] = e0 / s; q.p[1] = e1 / s; q.p[2] = e2 / s;
    return q;
}

// gradient of the softmax: g_logit_i = p_i * (g_i - sum_j g_j p_j)
__device__ __forceinline__ void softmax_bwd(const Point& q, const float gp[3], float gl[3]) {
    const float dot = gp[0] * q.p[0] + gp[1] * q.p[1] + gp[2] * q.p[2];
#pragma unroll
    for (int i = 0; i < 3; i++) gl[i] = q.p[i] * (gp[i] - dot);
}

__device__ __forceinline__ void load8(const float* __restrict__ raw, long row, float r[8]) {
    const float4 a = *reinterpret_cast<const float4*>(raw + 8 * row), b = *reinterpret_cast<const float4*>(raw + 8 * row + 4);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
}

__device__ __forceinline__ void store8(float* __restrict__ g, long row, const float v[8]) {
    *reinterpret_cast<float4*>(g + 8 * row) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(g + 8 * row + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void decode_weights_fwd_kernel(liso_slim_decode_cfg c, const float* __restrict__ raw,
                                                                      const int* __restrict__ lin, const uint8_t* __restrict__ filled,
                                                                      const float* __restrict__ ext, const float* __restrict__ pc,
                                                                      int pcs, float* __restrict__ x, float* __restrict__ y,
                                                                      float* __restrict__ w) {
    const long row = (long)blockIdx.x * kThreads + threadIdx.x;
    if (row >= (long)c.samples * c.n) return;
    const int cell = lin[row];
    if (cell < 0) {
        x[3 * row] = x[3 * row + 1] = x[3 * row + 2] = 0.f;
        y[3 * row] = y[3 * row + 1] = y[3 * row + 2] = 0.f;
        w[row] = 0.f;
        return;
    }
    const Consts k = logit_consts(c, ext);
    float r[8];
    load8(raw, row, r);
    const bool f = filled[cell] != 0;
    const Point q = decode(c, k, r, f);
    const float px = pc[(size_t)row * pcs], py = pc[(size_t)row * pcs + 1], pz = pc[(size_t)row * pcs + 2];
    x[3 * row] = px; x[3 * row + 1] = py; x[3 * row + 2] = pz;
    y[3 * row] = px + q.sf[0]; y[3 * row + 1] = py + q.sf[1]; y[3 * row + 2] = pz + 0.f;
    w[row] = q.p[0] * (f ? 1.f : 0.f);
}

__global__ __launch_bounds__(kThreads) void decode_weights_bwd_kernel(liso_slim_decode_cfg c, const float* __restrict__ raw,
                                                                      const int* __restrict__ lin, const uint8_t* __restrict__ filled,
                                                                      const float* __restrict__ ext, const float* __restrict__ gy,
                                                                      const float* __restrict__ gw, float* __restrict__ graw) {
    const long row = (long)blockIdx.x * kThreads + threadIdx.x;
    if (row >= (long)c.samples * c.n) return;
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int cell = lin[row];
    if (cell >= 0) {
        const Consts k = logit_consts(c, ext);
        float r[8];
        load8(raw, row, r);
        const bool f = filled[cell] != 0;
        const Point q = decode(c, k, r, f);
        if (gy && q.live_sf) { g[4] = gy[3 * row]; g[5] = gy[3 * row + 1]; }
        if (gw && f) {
            const float gp[3] = {gw[row], 0.f, 0.f};
            float gl[3];
            softmax_bwd(q, gp, gl);
#pragma unroll
            for (int i = 0; i < 3; i++) g[1 + i] = q.live_logit[1 + i] ? gl[i] : 0.f;
        }
    }
    store8(graw, row, g);
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------------------------
struct Sel {
    bool is_static, is_dynamic, is_ground;
    float saf[2];  // rigid flow of the pillar centre under the static-aggregation transform (unmasked)
};

__device__ __forceinline__ Sel select(const liso_slim_decode_cfg& c, const Point& q, float thr, const double* __restrict__ T, int cell) {
    Sel s;
    s.is_dynamic = q.p[1] >= thr;
    s.is_static = (q.p[0] >= q.p[2]) && !s.is_dynamic;
    s.is_ground = !(s.is_static || s.is_dynamic);
    const int hw = c.h * c.w;
    const int rc = cell % hw, rr = rc / c.w, cc = rc % c.w;
    const double cx = (((double)rr + 0.5) / (double)c.h) * c.ext_span[0] + c.ext_lo[0];
    const double cy = (((double)cc + 0.5) / (double)c.w) * c.ext_span[1] + c.ext_lo[1];
    // (T - I) applied to (cx, cy, 0, 1): head_decoder / static_aggregation.py:88-99 in fp64, then fp32
    s.saf[0] = (float)((T[0] - 1.0) * cx + T[1] * cy + T[3]);
    s.saf[1] = (float)(T[4] * cx + (T[5] - 1.0) * cy + T[7]);
    return s;
}

__global__ __launch_bounds__(kThreads) void decode_points_fwd_kernel(liso_slim_decode_cfg c, const float* __restrict__ raw,
                                                                     const int* __restrict__ lin, const uint8_t* __restrict__ filled,
                                                                     const float* __restrict__ ext, const float* __restrict__ thr_p,
                                                                     const double* __restrict__ trafo, liso_slim_decode_out o) {
    const long row = (long)blockIdx.x * kThreads + threadIdx.x;
    if (row >= (long)c.samples * c.n) return;
    const int cell = lin[row];
    float dis_l = 0.f, dis = 0.f, lg[3] = {0.f, 0.f, 0.f}, p[3] = {0.f, 0.f, 0.f}, dyn[2] = {0.f, 0.f}, st[2] = {0.f, 0.f},
          ag[2] = {0.f, 0.f}, sa[2] = {0.f, 0.f};
    uint8_t fl[3] = {0, 0, 0};
    if (cell >= 0) {
        const Consts k = logit_consts(c, ext);
        float r[8];
        load8(raw, row, r);
        const bool f = filled[cell] != 0;
        const Point q = decode(c, k, r, f);
        const Sel s = select(c, q, thr_p[0], trafo + (size_t)(row / c.n) * 16, cell);
        dis_l = q.logit[0];
        dis = 1.f / (1.f + expf(-dis_l));
#pragma unroll
        for (int i = 0; i < 3; i++) { lg[i] = q.logit[1 + i]; p[i] = q.p[i]; }
        dyn[0] = q.df[0]; dyn[1] = q.df[1];
        st[0] = q.sf[0]; st[1] = q.sf[1];
        sa[0] = s.saf[0]; sa[1] = s.saf[1];
        const float one_m_g = 1.f - q.p[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float static2 = c.use_static_aggr ? (f ? s.saf[i] : 0.f) : q.sf[i];
            float dyn2 = q.df[i] * one_m_g;
            if (c.non_rigid) dyn2 = static2 * one_m_g + dyn2;
            ag[i] = s.is_static ? static2 : dyn2;
        }
        fl[0] = s.is_static; fl[1] = s.is_dynamic; fl[2] = s.is_ground;
    }
    if (o.dis_logit) o.dis_logit[row] = dis_l;
    if (o.dis) o.dis[row] = dis;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (o.logits) o.logits[3 * row + i] = lg[i];
        if (o.probs) o.probs[3 * row + i] = p[i];
        if (o.flags) o.flags[3 * row + i] = fl[i];
    }
    if (o.staticness) o.staticness[row] = p[0];
    if (o.dynamicness) o.dynamicness[row] = p[1];
    if (o.groundness) o.groundness[row] = p[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (o.dyn_flow) o.dyn_flow[3 * row + i] = i < 2 ? dyn[i] : 0.f;
        if (o.stat_flow) o.stat_flow[3 * row + i] = i < 2 ? st[i] : 0.f;
        if (o.agg_flow) o.agg_flow[3 * row + i] = i < 2 ? ag[i] : 0.f;
        if (o.saf_flow) o.saf_flow[3 * row + i] = i < 2 ? sa[i] : 0.f;
    }
}

__global__ __launch_bounds__(kThreads) void decode_points_bwd_kernel(liso_slim_decode_cfg c, const float* __restrict__ raw,
                                                                     const int* __restrict__ lin, const uint8_t* __restrict__ filled,
                                                                     const float* __restrict__ ext, const float* __restrict__ thr_p,
                                                                     const double* __restrict__ trafo, liso_slim_decode_out go,
                                                                     float* __restrict__ graw, float* __restrict__ gsaf) {
    const long row = (long)blockIdx.x * kThreads + threadIdx.x;
    if (row >= (long)c.samples * c.n) return;
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gs[2] = {0.f, 0.f};
    const int cell = lin[row];
    if (cell >= 0) {
        const Consts k = logit_consts(c, ext);
        float r[8];
        load8(raw, row, r);
        const bool f = filled[cell] != 0;
        const Point q = decode(c, k, r, f);
        const Sel s = select(c, q, thr_p[0], trafo + (size_t)(row / c.n) * 16, cell);
        float gl[4] = {0.f, 0.f, 0.f, 0.f};   // d / d logit (post modes): disappearing, static, dynamic, ground
        float gp[3] = {0.f, 0.f, 0.f};        // d / d class probability
        float gsf[2] = {0.f, 0.f}, gdf[2] = {0.f, 0.f};
        if (go.dis_logit) gl[0] += go.dis_logit[row];
        if (go.dis) {
            const float d = 1.f / (1.f + expf(-q.logit[0]));
            gl[0] += go.dis[row] * d * (1.f - d);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            if (go.logits) gl[1 + i] += go.logits[3 * row + i];
            if (go.probs) gp[i] += go.probs[3 * row + i];
        }
        if (go.staticness) gp[0] += go.staticness[row];
        if (go.dynamicness) gp[1] += go.dynamicness[row];
        if (go.groundness) gp[2] += go.groundness[row];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            if (go.dyn_flow) gdf[i] += go.dyn_flow[3 * row + i];
            if (go.stat_flow) gsf[i] += go.stat_flow[3 * row + i];
            if (go.saf_flow) gs[i] += go.saf_flow[3 * row + i];
            if (go.agg_flow) {
                const float ga = go.agg_flow[3 * row + i];
                const float one_m_g = 1.f - q.p[2];
                const float static2 = c.use_static_aggr ? (f ? s.saf[i] : 0.f) : q.sf[i];
                float g_static2 = 0.f;
                if (s.is_static) {
                    g_static2 = ga;
                } else {
                    gdf[i] += ga * one_m_g;
                    gp[2] -= ga * q.df[i];
                    if (c.non_rigid) {
                        g_static2 = ga * one_m_g;
                        gp[2] -= ga * static2;
                    }
                }
                if (c.use_static_aggr) { if (f) gs[i] += g_static2; }
                else gsf[i] += g_static2;
            }
        }
        float gls[3];
        softmax_bwd(q, gp, gls);
#pragma unroll
        for (int i = 0; i < 3; i++) gl[1 + i] += gls[i];
#pragma unroll
        for (int i = 0; i < 4; i++) g[i] = q.live_logit[i] ? gl[i] : 0.f;
        if (q.live_sf) { g[4] = gsf[0]; g[5] = gsf[1]; }
        if (q.live_df) { g[6] = gdf[0] * c.dyn_grad_scale; g[7] = gdf[1] * c.dyn_grad_scale; }
    }
    store8(graw, row, g);
    if (gsaf) { gsaf[2 * row] = gs[0]; gsaf[2 * row + 1] = gs[1]; }
}

// ---- masked means ------------------------------------------------------------------------------------------------------------------
// workspace: double partial[2 * kRedBlocks] | double total[2] (sum, count)
__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void block_partial(double sum, double cnt, double* __restrict__ partial) {
    __shared__ double red[2][kThreads / 64];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { sum += shfl_xor_f64(sum, m); cnt += shfl_xor_f64(cnt, m); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = sum; red[1][wave] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0, n = 0.0;
        for (int i = 0; i < kThreads / 64; i++) { s += red[0][i]; n += red[1][i]; }
        partial[2 * blockIdx.x] = s;
        partial[2 * blockIdx.x + 1] = n;
    }
}

__global__ void masked_mean_final_kernel(double* __restrict__ ws, float* __restrict__ out) {
    // one wave, fixed order
    double s = 0.0, n = 0.0;
    for (int i = threadIdx.x; i < kRedBlocks; i += 64) { s += ws[2 * i]; n += ws[2 * i + 1]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s += shfl_xor_f64(s, m); n += shfl_xor_f64(n, m); }
    if (threadIdx.x == 0) {
        ws[2 * kRedBlocks] = s;
        ws[2 * kRedBlocks + 1] = n;
        out[0] = (float)s / (float)n;  // torch: fp32 sum / fp32 count (0 / 0 = NaN as there)
    }
}

// static_points_loss (slim_loss_adaptor.py:55-91) of one row: est - flow per component, fp64 transform
__device__ __forceinline__ void static_residual(const float* __restrict__ pc, int pcs, long row, const double* __restrict__ T,
                                                const float* __restrict__ flow, float d[3]) {
    const float px = pc[(size_t)row * pcs], py = pc[(size_t)row * pcs + 1], pz = pc[(size_t)row * pcs + 2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double moved = (double)px * T[4 * i] + (double)py * T[4 * i + 1] + (double)pz * T[4 * i + 2] + T[4 * i + 3];
        const float pi = i == 0 ? px : (i == 1 ? py : pz);
        const float est = (float)(moved - (double)pi);
        d[i] = est - flow[3 * row + i];
    }
}

__global__ __launch_bounds__(kThreads) void static_loss_fwd_kernel(int samples, long n, const float* __restrict__ pc, int pcs,
                                                                   const uint8_t* __restrict__ valid, const float* __restrict__ flow,
                                                                   const float* __restrict__ weight, const double* __restrict__ trafo,
                                                                   double* __restrict__ partial) {
    double sum = 0.0, cnt = 0.0;
    const long rows = (long)samples * n;
    for (long row = (long)blockIdx.x * kThreads + threadIdx.x; row < rows; row += (long)kRedBlocks * kThreads) {
        if (!valid[row]) continue;
        float d[3];
        static_residual(pc, pcs, row, trafo + (size_t)(row / n) * 16, flow, d);
        const float w = weight[row];
        const float l = (w * (d[0] * d[0]) + w * (d[1] * d[1]) + w * (d[2] * d[2])) / 3.f;
        sum += (double)l;
        cnt += 1.0;
    }
    block_partial(sum, cnt, partial);
}

__global__ __launch_bounds__(kThreads) void static_loss_bwd_kernel(int samples, long n, const float* __restrict__ pc, int pcs,
                                                                   const uint8_t* __restrict__ valid, const float* __restrict__ flow,
                                                                   const float* __restrict__ weight, const double* __restrict__ trafo,
                                                                   const float* __restrict__ gout, const double* __restrict__ ws,
                                                                   float* __restrict__ gflow, float* __restrict__ gweight) {
    const long row = (long)blockIdx.x * kThreads + threadIdx.x;
    if (row >= (long)samples * n) return;
    float gf[3] = {0.f, 0.f, 0.f}, gw = 0.f;
    if (valid[row]) {
        const float scale = gout[0] / (float)ws[2 * kRedBlocks + 1] / 3.f;
        float d[3];
        static_residual(pc, pcs, row, trafo + (size_t)(row / n) * 16, flow, d);
        const float w = weight[row];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            gf[i] = scale * w * -2.f * d[i];   // d = est - flow
            gw += scale * d[i] * d[i];
        }
    }
    if (gflow) { gflow[3 * row] = gf[0]; gflow[3 * row + 1] = gf[1]; gflow[3 * row + 2] = gf[2]; }
    if (gweight) gweight[row] = gw;
}

// ---- nearest-point loss -------------------------------------------------------------------------------------------------------------
struct FlowPtrs {
    const float* p[8];
};

__global__ __launch_bounds__(kThreads) void knn_queries_kernel(int samples, int clouds, long n, int types, const float* __restrict__ pc, int pcs,
                                                               const uint8_t* __restrict__ valid, FlowPtrs flows,
                                                               const long long* __restrict__ order, float* __restrict__ query) {
    const long j = (long)blockIdx.x * kThreads + threadIdx.x;  // position in the query order
    const int s = blockIdx.y;
    if (j >= n) return;
    const long i = order ? (long)order[(size_t)(s % clouds) * n + j] : j;
    const long row = (long)s * n + i;
    const bool ok = valid[row] != 0;
    const float nanv = nanf("");
    const float px = pc[(size_t)row * pcs], py = pc[(size_t)row * pcs + 1], pz = pc[(size_t)row * pcs + 2];
    for (int t = 0; t < types; t++) {
        const float* f = flows.p[t] + 3 * row;
        float* q = query + (((size_t)t * samples + s) * n + j) * 3;
        q[0] = ok ? px + f[0] : nanv;
        q[1] = ok ? py + f[1] : nanv;
        q[2] = ok ? pz + f[2] : nanv;
    }
}

struct NpEval {
    float d2, loss, dx, dy, dz, scale;
    bool ok;
};

// knn_wrapper.py:58-135 + huber_delta (:11-49) for one valid row (same arithmetic as slim_loss.hip)
__device__ __forceinline__ NpEval np_eval(const liso_slim_nploss_cfg& c, float qx, float qy, float qz, const float* __restrict__ nn) {
    NpEval e;
    e.dx = nn[0] - qx; e.dy = nn[1] - qy; e.dz = nn[2] - qz;
    e.d2 = e.dx * e.dx + e.dy * e.dy + e.dz * e.dz;
    const float min_fov = fminf(fminf(qx - c.ext[0], qy - c.ext[1]), fminf(c.ext[2] - qx, c.ext[3] - qy));
    float w = 1.f;
    if (c.fov_mode == 1) w = min_fov > 0.f ? 1.f : 0.f;
    else if (c.fov_mode == 2) w = (min_fov > 0.f && e.d2 < min_fov * min_fov) ? 1.f : 0.f;
    float l, dl;
    if (c.delta == 0.f) {
        const bool nz = !(e.d2 == 0.f);
        l = nz ? sqrtf(e.d2) : 0.f;
        dl = nz ? 0.5f / l : 0.f;
    } else {
        const float dd = c.delta * c.delta;
        l = fminf(e.d2, dd) / (2.f * c.delta) + sqrtf(fmaxf(e.d2, dd)) - c.delta;
        dl = e.d2 < dd ? 1.f / (2.f * c.delta) : (e.d2 > dd ? 0.5f / sqrtf(e.d2) : 1.f / c.delta);
    }
    e.loss = l * w;
    e.scale = -2.f * dl * w;
    e.ok = isfinite(e.d2);
    return e;
}

__global__ __launch_bounds__(kThreads) void nploss_masked_fwd_kernel(liso_slim_nploss_cfg c, const float* __restrict__ pc, int pcs,
                                                                     const uint8_t* __restrict__ valid, const float* __restrict__ flow,
                                                                     const float* __restrict__ cloud_b, int cbs,
                                                                     const long long* __restrict__ index,
                                                                     const long long* __restrict__ order, float* __restrict__ dist_sqr,
                                                                     double* __restrict__ partial) {
    double sum = 0.0, cnt = 0.0;
    const long rows = (long)c.samples * c.n;
    for (long qrow = (long)blockIdx.x * kThreads + threadIdx.x; qrow < rows; qrow += (long)kRedBlocks * kThreads) {
        const long s = qrow / c.n, j = qrow - s * c.n;
        const long i = order ? (long)order[(size_t)(s % c.clouds) * c.n + j] : j;
        const long row = s * c.n + i;
        if (!valid[row]) { dist_sqr[row] = 0.f; continue; }
        const float qx = pc[(size_t)row * pcs] + flow[3 * row], qy = pc[(size_t)row * pcs + 1] + flow[3 * row + 1],
                    qz = pc[(size_t)row * pcs + 2] + flow[3 * row + 2];
        const long nb = (long)index[qrow];
        const NpEval e = np_eval(c, qx, qy, qz, cloud_b + ((size_t)s * c.n_b + (nb >= 0 && nb < c.n_b ? nb : 0)) * cbs);
        dist_sqr[row] = e.d2;
        sum += (double)e.loss;
        cnt += 1.0;
    }
    block_partial(sum, cnt, partial);
}

__global__ __launch_bounds__(kThreads) void nploss_masked_bwd_kernel(liso_slim_nploss_cfg c, const float* __restrict__ pc, int pcs,
                                                                     const uint8_t* __restrict__ valid, const float* __restrict__ flow,
                                                                     const float* __restrict__ cloud_b, int cbs,
                                                                     const long long* __restrict__ index,
                                                                     const long long* __restrict__ order, const float* __restrict__ gout,
                                                                     const float* __restrict__ gd2, const double* __restrict__ ws,
                                                                     float* __restrict__ gflow) {
    const long qrow = (long)blockIdx.x * kThreads + threadIdx.x;
    if (qrow >= (long)c.samples * c.n) return;
    const long s = qrow / c.n, j = qrow - s * c.n;
    const long i = order ? (long)order[(size_t)(s % c.clouds) * c.n + j] : j;
    const long row = s * c.n + i;
    float g[3] = {0.f, 0.f, 0.f};
    if (valid[row]) {
        const float qx = pc[(size_t)row * pcs] + flow[3 * row], qy = pc[(size_t)row * pcs + 1] + flow[3 * row + 1],
                    qz = pc[(size_t)row * pcs + 2] + flow[3 * row + 2];
        const long nb = (long)index[qrow];
        const NpEval e = np_eval(c, qx, qy, qz, cloud_b + ((size_t)s * c.n_b + (nb >= 0 && nb < c.n_b ? nb : 0)) * cbs);
        float sc = gout[0] / (float)ws[2 * kRedBlocks + 1] * e.scale;
        if (gd2) sc += gd2[row] * -2.f;
        if (e.ok) { g[0] = sc * e.dx; g[1] = sc * e.dy; g[2] = sc * e.dz; }
    }
    gflow[3 * row] = g[0]; gflow[3 * row + 1] = g[1]; gflow[3 * row + 2] = g[2];
}

// ---- per-channel extrema of a [rows, stride] map's first c <= 4 channels (the decoder's global logit extrema, head_decoder.py:779-955) ----
// max / min are exact in any order: block partials, one final block.  NaN propagates like torch.amax / amin.
constexpr int kExtBlocks = 512;
__device__ __forceinline__ float nan_max(float m, float v) { return (v > m || v != v) ? v : m; }
__device__ __forceinline__ float nan_min(float m, float v) { return (v < m || v != v) ? v : m; }

__global__ __launch_bounds__(256) void channel_extrema_partial_kernel(const float* __restrict__ x, long rows, int stride, int c,
                                                                      float* __restrict__ partial) {
    float hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { hi[k] = -INFINITY; lo[k] = INFINITY; }
    const bool vec = c == 4 && (stride & 3) == 0 && (((uintptr_t)x) & 15) == 0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
        float v[4];
        if (vec) {
            const float4 q = *reinterpret_cast<const float4*>(x + r * stride);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = k < c ? x[r * stride + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { hi[k] = nan_max(hi[k], v[k]); lo[k] = nan_min(lo[k], v[k]); }
    }
    __shared__ float red[4][8];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { hi[k] = nan_max(hi[k], __shfl_xor(hi[k], o)); lo[k] = nan_min(lo[k], __shfl_xor(lo[k], o)); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 4; k++) { red[wave][k] = hi[k]; red[wave][4 + k] = lo[k]; }
    __syncthreads();
    if (threadIdx.x < 8) {
        const bool is_hi = threadIdx.x < 4;
        float m = red[0][threadIdx.x];
        for (int w = 1; w < 4; w++) m = is_hi ? nan_max(m, red[w][threadIdx.x]) : nan_min(m, red[w][threadIdx.x]);
        partial[(long)blockIdx.x * 8 + threadIdx.x] = m;
    }
}

__global__ __launch_bounds__(256) void channel_extrema_final_kernel(const float* __restrict__ partial, int nblk, int c, float* __restrict__ out) {
    __shared__ float red[32][8];
    const int col = threadIdx.x & 7, grp = threadIdx.x >> 3;  // 32 groups x 8 columns (4 maxima | 4 minima)
    const bool is_hi = col < 4;
    float m = is_hi ? -INFINITY : INFINITY;
    for (int b = grp; b < nblk; b += 32) m = is_hi ? nan_max(m, partial[(long)b * 8 + col]) : nan_min(m, partial[(long)b * 8 + col]);
    red[grp][col] = m;
    __syncthreads();
    if (threadIdx.x < 8) {
        for (int g = 1; g < 32; g++) m = is_hi ? nan_max(m, red[g][col]) : nan_min(m, red[g][col]);
        if ((col & 3) < c) out[(is_hi ? 0 : c) + (col & 3)] = m;
    }
}

inline bool cfg_ok(const liso_slim_decode_cfg* c) {
    if (!c || c->samples < 1 || c->n < 0 || c->h < 1 || c->w < 1) return false;
    for (int i = 0; i < 4; i++)
        if (c->logit_mode[i] < 0 || c->logit_mode[i] > 2) return false;
    return (long)c->samples * c->h * c->w < (1L << 31);
}

inline unsigned blocks_for(long rows) { return (unsigned)((rows + kThreads - 1) / kThreads); }

}  // namespace

extern "C" {

int liso_slim_decode_weights_fwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                 const float* extrema, const float* pc, int pc_stride, float* x, float* y, float* w, void* stream) {
    if (!cfg_ok(cfg) || pc_stride < 3) return LISO_EINVAL;
    const long rows = (long)cfg->samples * cfg->n;
    if (rows == 0) return LISO_OK;
    if (!raw || !lin || !filled || !pc || !x || !y || !w) return LISO_EINVAL;
    decode_weights_fwd_kernel<<<blocks_for(rows), kThreads, 0, (hipStream_t)stream>>>(*cfg, raw, lin, filled, extrema, pc, pc_stride, x,
                                                                                      y, w);
    return check_launch();
}

int liso_slim_decode_weights_bwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                 const float* extrema, const float* grad_y, const float* grad_w, float* grad_raw, void* stream) {
    if (!cfg_ok(cfg)) return LISO_EINVAL;
    const long rows = (long)cfg->samples * cfg->n;
    if (rows == 0) return LISO_OK;
    if (!raw || !lin || !filled || !grad_raw) return LISO_EINVAL;
    decode_weights_bwd_kernel<<<blocks_for(rows), kThreads, 0, (hipStream_t)stream>>>(*cfg, raw, lin, filled, extrema, grad_y, grad_w,
                                                                                      grad_raw);
    return check_launch();
}

int liso_slim_decode_points_fwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                const float* extrema, const float* threshold, const double* trafo, const liso_slim_decode_out* out,
                                void* stream) {
    if (!cfg_ok(cfg) || !out) return LISO_EINVAL;
    const long rows = (long)cfg->samples * cfg->n;
    if (rows == 0) return LISO_OK;
    if (!raw || !lin || !filled || !threshold || !trafo) return LISO_EINVAL;
    decode_points_fwd_kernel<<<blocks_for(rows), kThreads, 0, (hipStream_t)stream>>>(*cfg, raw, lin, filled, extrema, threshold, trafo,
                                                                                     *out);
    return check_launch();
}

int liso_slim_decode_points_bwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                const float* extrema, const float* threshold, const double* trafo, const liso_slim_decode_out* grad,
                                float* grad_raw, float* grad_saf_eff, void* stream) {
    if (!cfg_ok(cfg) || !grad) return LISO_EINVAL;
    const long rows = (long)cfg->samples * cfg->n;
    if (rows == 0) return LISO_OK;
    if (!raw || !lin || !filled || !threshold || !trafo || !grad_raw) return LISO_EINVAL;
    decode_points_bwd_kernel<<<blocks_for(rows), kThreads, 0, (hipStream_t)stream>>>(*cfg, raw, lin, filled, extrema, threshold, trafo,
                                                                                     *grad, grad_raw, grad_saf_eff);
    return check_launch();
}

size_t liso_slim_loss_workspace_bytes(void) { return (size_t)(2 * kRedBlocks + 2) * sizeof(double); }

int liso_slim_static_points_loss_fwd(int samples, long n, const float* pc, int pc_stride, const uint8_t* valid, const float* flow,
                                     const float* weight, const double* trafo, float* out, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (samples < 1 || n < 0 || pc_stride < 3 || !out || !workspace || workspace_bytes < liso_slim_loss_workspace_bytes())
        return LISO_EINVAL;
    if (n > 0 && (!pc || !valid || !flow || !weight || !trafo)) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    static_loss_fwd_kernel<<<kRedBlocks, kThreads, 0, st>>>(samples, n, pc, pc_stride, valid, flow, weight, trafo, (double*)workspace);
    masked_mean_final_kernel<<<1, 64, 0, st>>>((double*)workspace, out);
    return check_launch();
}

int liso_slim_static_points_loss_bwd(int samples, long n, const float* pc, int pc_stride, const uint8_t* valid, const float* flow,
                                     const float* weight, const double* trafo, const float* grad_out, const void* workspace,
                                     float* grad_flow, float* grad_weight, void* stream) {
    if (samples < 1 || n < 0 || pc_stride < 3 || !grad_out || !workspace) return LISO_EINVAL;
    const long rows = (long)samples * n;
    if (rows == 0) return LISO_OK;
    if (!pc || !valid || !flow || !weight || !trafo) return LISO_EINVAL;
    static_loss_bwd_kernel<<<blocks_for(rows), kThreads, 0, (hipStream_t)stream>>>(samples, n, pc, pc_stride, valid, flow, weight, trafo,
                                                                                   grad_out, (const double*)workspace, grad_flow,
                                                                                   grad_weight);
    return check_launch();
}

int liso_slim_knn_queries(int samples, int clouds, long n, int types, const float* pc, int pc_stride, const uint8_t* valid,
                          const float* const* flows, const int64_t* order, float* query, void* stream) {
    if (samples < 1 || clouds < 1 || n < 0 || types < 1 || types > 8 || pc_stride < 3 || !flows) return LISO_EINVAL;
    if (n == 0) return LISO_OK;
    if (!pc || !valid || !query) return LISO_EINVAL;
    FlowPtrs fp;
    for (int t = 0; t < 8; t++) fp.p[t] = t < types ? flows[t] : nullptr;
    for (int t = 0; t < types; t++)
        if (!fp.p[t]) return LISO_EINVAL;
    dim3 grid(blocks_for(n), (unsigned)samples);
    knn_queries_kernel<<<grid, kThreads, 0, (hipStream_t)stream>>>(samples, clouds, n, types, pc, pc_stride, valid, fp, (const long long*)order,
                                                                   query);
    return check_launch();
}

static bool np_ok(const liso_slim_nploss_cfg* c) {
    return c && c->samples >= 1 && c->clouds >= 1 && c->n >= 0 && c->n_b >= 1 && c->fov_mode >= 0 && c->fov_mode <= 2 && c->delta >= 0.f;
}

int liso_slim_nearest_point_loss_fwd(const liso_slim_nploss_cfg* cfg, const float* pc, int pc_stride, const uint8_t* valid,
                                     const float* flow, const float* cloud_b, int cloud_b_stride, const int64_t* index,
                                     const int64_t* order, float* dist_sqr, float* out, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    if (!np_ok(cfg) || pc_stride < 3 || cloud_b_stride < 3 || !out || !workspace || workspace_bytes < liso_slim_loss_workspace_bytes())
        return LISO_EINVAL;
    if (cfg->n > 0 && (!pc || !valid || !flow || !cloud_b || !index || !dist_sqr)) return LISO_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    nploss_masked_fwd_kernel<<<kRedBlocks, kThreads, 0, st>>>(*cfg, pc, pc_stride, valid, flow, cloud_b, cloud_b_stride,
                                                              (const long long*)index, (const long long*)order, dist_sqr,
                                                              (double*)workspace);
    masked_mean_final_kernel<<<1, 64, 0, st>>>((double*)workspace, out);
    return check_launch();
}

int liso_slim_nearest_point_loss_bwd(const liso_slim_nploss_cfg* cfg, const float* pc, int pc_stride, const uint8_t* valid,
                                     const float* flow, const float* cloud_b, int cloud_b_stride, const int64_t* index,
                                     const int64_t* order, const float* grad_out, const float* grad_dist_sqr, const void* workspace,
                                     float* grad_flow, void* stream) {
    if (!np_ok(cfg) || pc_stride < 3 || cloud_b_stride < 3 || !grad_out || !workspace) return LISO_EINVAL;
    const long rows = (long)cfg->samples * cfg->n;
    if (rows == 0) return LISO_OK;
    if (!pc || !valid || !flow || !cloud_b || !index || !grad_flow) return LISO_EINVAL;
    nploss_masked_bwd_kernel<<<blocks_for(rows), kThreads, 0, (hipStream_t)stream>>>(*cfg, pc, pc_stride, valid, flow, cloud_b,
                                                                                    cloud_b_stride, (const long long*)index,
                                                                                    (const long long*)order, grad_out, grad_dist_sqr,
                                                                                    (const double*)workspace, grad_flow);
    return check_launch();
}


size_t liso_channel_extrema_workspace_bytes(void) { return (size_t)kExtBlocks * 8 * sizeof(float); }

int liso_channel_extrema_f32(const float* x, long rows, int stride, int c, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (rows <= 0 || c < 1 || c > 4 || stride < c) return LISO_EINVAL;
    if (!x || !out || !workspace) return LISO_EINVAL;
    if (workspace_bytes < liso_channel_extrema_workspace_bytes()) return LISO_EWORKSPACE;
    long nb = (rows + 255) / 256;
    nb = nb > kExtBlocks ? kExtBlocks : nb;
    hipStream_t st = (hipStream_t)stream;
    channel_extrema_partial_kernel<<<(unsigned)nb, 256, 0, st>>>(x, rows, stride, c, (float*)workspace);
    channel_extrema_final_kernel<<<1, 256, 0, st>>>((const float*)workspace, (int)nb, c, out);
    return hipGetLastError() == hipSuccess ? LISO_OK : LISO_ELAUNCH;
}

}  // extern "C"
