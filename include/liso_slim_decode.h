/*
 * liso_slim_decode.h -- C ABI of the per-point SLIM training decoder and of its point-wise loss terms (gfx950).
 *
 * What the reference does with ~250 framework launches forward and ~150 backward per training step on [12, N, C] tensors
 * (one per RAFT iteration and flow direction):
 *   liso/slim/model/head_decoder.py:67-298   apply_output_modification: channel split, artificial flows / logits (:734-955),
 *                                            defaults at unfilled pillars (:566-590), class softmax, dynamicness threshold,
 *                                            static / dynamic / ground decision (:592-612), flow selection (:236-283)
 *   liso/slim/model/head_decoder.py:300-408  apply_flow_to_points (values of every map at the point's pillar)
 *   liso/slim/slim_loss/static_aggregation.py:34-110   weights + warped cloud of the static-flow Kabsch fit, rigid flow of the
 *                                            pillar centres under the fitted transform
 *   liso/slim/slim_loss/slim_loss_adaptor.py:55-91     static_points_loss (rigid flow of every point vs predicted static flow)
 *   liso/slim/slim_loss/knn_loss.py:9-82, knn_wrapper.py:155-217   NaN-marking of padding rows, warped query cloud, masked mean
 * Every step is point-wise once the 8 raw channels of the point's pillar are known (liso_bev_gather_fwd_f32 of liso_slim.h),
 * except the Kabsch fit itself (weighted moments + 3x3 solve, liso_kabsch.h), which sits between the two decode passes:
 *
 *   raw [S,N,8] --decode_weights--> x, y, w --moments/solve--> T [S,4,4] --decode_points--> per-point predictions --losses
 *
 * All arrays are dense and row-major; S = samples (RAFT iterations x directions x batch), N = padded points per sample.
 * Invalid (padding) rows produce zeros in every output and zero gradients.  Nothing allocates or synchronises; every function
 * returns 0 or a negative LISO_E* code (liso_iou3d.h).
 */
#ifndef LISO_SLIM_DECODE_H
#define LISO_SLIM_DECODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISO_DECODE_NET 0   /* use the network's channel */
#define LISO_DECODE_ON 1    /* output_modification `True`: logit forced on (max of the other two + 100, :812-817 etc.); flows: "zero" */
#define LISO_DECODE_OFF 2   /* output_modification `False`: logit forced off */

typedef struct {
    int samples;
    long n;
    int h, w;                /* BEV grid of the network output (pillar grid / final_scale) */
    int logit_mode[4];       /* disappearing, static, dynamic, ground: LISO_DECODE_NET / ON / OFF */
    int static_flow_zero;    /* output_modification.static_flow == "zero" */
    int dynamic_flow_zero;   /* output_modification.dynamic_flow == "zero" */
    int overwrite_flow;      /* overwrite_non_filled_pillars_with_default_flow */
    int overwrite_logits;    /* overwrite_non_filled_pillars_with_default_logits */
    int non_rigid;           /* model.dynamic_flow_is_non_rigid_flow */
    int use_static_aggr;     /* model.use_static_aggr_flow_for_aggr_flow */
    float dyn_grad_scale;    /* output_modification.dynamic_flow_grad_scale (backward only) */
    double ext_lo[2], ext_span[2]; /* BEV extent: centre of cell (r, c) = ((r + 0.5) / h) * span[0] + lo[0], ... (bev_utils.py:24-40) */
} liso_slim_decode_cfg;

/* Inputs shared by the calls below:
 *   raw     float32 [S,N,8]   network output at every point's pillar: 4 logits | static flow xy | dynamic flow xy (0 at invalid rows)
 *   lin     int32 [S*N]       flattened cell (s*h*w + r*w + c) of every point, < 0 = invalid row
 *   filled  uint8 [S*h*w]     filled-pillar mask
 *   extrema float32 [8]       max[4] | min[4] of the raw logit channels over the whole BEV batch (read only if a class logit is ON / OFF)
 */

/* Pass 1 (static_aggregation.py:34-68): x = point (0 at invalid rows), y = x + (static flow, 0), w = staticness * filled.
 *   pc float32 [S,N,pc_stride] (xyz first).  Backward: grad_y [S,N,3], grad_w [S,N] (either may be NULL) -> grad_raw [S,N,8]. */
int liso_slim_decode_weights_fwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                 const float* extrema, const float* pc, int pc_stride, float* x, float* y, float* w, void* stream);
int liso_slim_decode_weights_bwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                 const float* extrema, const float* grad_y, const float* grad_w, float* grad_raw, void* stream);

/* Pass 2: every per-point prediction.  threshold: float32 [1] (device).  trafo float64 [S,4,4] (static aggregation).
 *   outputs (all [S,N,...], float32 unless noted): dis_logit [S,N], dis [S,N] = sigmoid, logits [S,N,3], probs [S,N,3],
 *   staticness / dynamicness / groundness [S,N] (= probs columns), dyn_flow / stat_flow / agg_flow / saf_flow [S,N,3] (z = 0),
 *   flags uint8 [S,N,3] = is_static, is_dynamic, is_ground. */
typedef struct {
    float *dis_logit, *dis, *logits, *probs, *staticness, *dynamicness, *groundness, *dyn_flow, *stat_flow, *agg_flow, *saf_flow;
    uint8_t* flags;
} liso_slim_decode_out;
int liso_slim_decode_points_fwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                const float* extrema, const float* threshold, const double* trafo, const liso_slim_decode_out* out,
                                void* stream);
/* Backward of pass 2: `grad` holds the gradients of the float outputs (any pointer may be NULL; flags ignored) -> grad_raw [S,N,8]
 * and, when grad_saf_eff != NULL, the total gradient reaching the rigid flow of the pillar centre [S,N,2] (from saf_flow and,
 * with use_static_aggr, from agg_flow) for the caller's reduction onto the transform. */
int liso_slim_decode_points_bwd(const liso_slim_decode_cfg* cfg, const float* raw, const int32_t* lin, const uint8_t* filled,
                                const float* extrema, const float* threshold, const double* trafo, const liso_slim_decode_out* grad,
                                float* grad_raw, float* grad_saf_eff, void* stream);

/* static_points_loss + masked mean (slim_loss_adaptor.py:55-91, :176-184): est = (T p)_xyz - p in fp64 -> fp32,
 * l = mean_c(weight * (est_c - flow_c)^2); out[0] = sum of l over valid rows / number of valid rows (0 rows -> NaN, as torch).
 *   valid uint8 [S,N]; flow [S,N,3]; weight [S,N]; trafo float64 [S,4,4]; workspace >= liso_slim_loss_workspace_bytes().
 * Backward: grad_out float32 [1] -> grad_flow [S,N,3], grad_weight [S,N] (either may be NULL). */
/* Per-channel maxima and minima of the first c <= 4 channels of a [rows, stride] fp32 map in one pass: out = [max_0..max_{c-1} |
 * min_0..min_{c-1}], NaN propagating like torch.amax / amin.  The decoder's global logit extrema of the True / False output modes
 * (liso/slim/model/head_decoder.py:779-955 takes torch.max / torch.min over the batch it is given): two launches instead of the two
 * staged framework reductions that read the strided logit channels twice. */
size_t liso_channel_extrema_workspace_bytes(void);
int liso_channel_extrema_f32(const float* x, long rows, int stride, int c, float* out, void* workspace, size_t workspace_bytes, void* stream);

size_t liso_slim_loss_workspace_bytes(void);
int liso_slim_static_points_loss_fwd(int samples, long n, const float* pc, int pc_stride, const uint8_t* valid, const float* flow,
                                     const float* weight, const double* trafo, float* out, void* workspace, size_t workspace_bytes,
                                     void* stream);
int liso_slim_static_points_loss_bwd(int samples, long n, const float* pc, int pc_stride, const uint8_t* valid, const float* flow,
                                     const float* weight, const double* trafo, const float* grad_out, const void* workspace,
                                     float* grad_flow, float* grad_weight, void* stream);

/* Warped query clouds of the nearest-point loss (knn_loss.py:44-58, knn_wrapper.py:170,186-190): for flow type t, sample s and
 * position j of the query order, query[t,s,j] = pc[s,o[j]] + flow_t[s,o[j]] (NaN at invalid rows) with o = order[s % clouds]
 * (samples are stacked [iteration][cloud]; every cloud has its own bucket order, slim_knn's sorted ids).
 *   flows: `types` <= 8 device pointers (host array) to float32 [S,N,3]; order int64 [clouds,N] or NULL (identity);
 *   query float32 [types,S,N,3]. */
int liso_slim_knn_queries(int samples, int clouds, long n, int types, const float* pc, int pc_stride, const uint8_t* valid,
                          const float* const* flows, const int64_t* order, float* query, void* stream);

/* Nearest-point loss with the padding mask, the query order and the masked mean in one pass (knn_wrapper.py:58-135,186-217 +
 * slim_loss_adaptor.py:239-251): index int64 [S,N] is in QUERY order (index[s,j] answers point order[s % clouds][j]).
 *   -> dist_sqr [S,N] in point order (0 at invalid rows), out[0] = sum of loss over valid rows / number of valid rows.
 * Backward: grad_out [1] (+ optional grad_dist_sqr [S,N]) -> grad_flow [S,N,3]. */
typedef struct {
    int samples, clouds;
    long n, n_b;
    float ext[4];
    int fov_mode;   /* as liso_nploss_cfg (liso_slim.h) */
    float delta;
} liso_slim_nploss_cfg;
int liso_slim_nearest_point_loss_fwd(const liso_slim_nploss_cfg* cfg, const float* pc, int pc_stride, const uint8_t* valid,
                                     const float* flow, const float* cloud_b, int cloud_b_stride, const int64_t* index,
                                     const int64_t* order, float* dist_sqr, float* out, void* workspace, size_t workspace_bytes,
                                     void* stream);
int liso_slim_nearest_point_loss_bwd(const liso_slim_nploss_cfg* cfg, const float* pc, int pc_stride, const uint8_t* valid,
                                     const float* flow, const float* cloud_b, int cloud_b_stride, const int64_t* index,
                                     const int64_t* order, const float* grad_out, const float* grad_dist_sqr, const void* workspace,
                                     float* grad_flow, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_SLIM_DECODE_H */
