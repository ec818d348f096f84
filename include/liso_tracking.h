/*
 * liso_tracking.h -- C ABI of the MI355X-native inner loops of box mining / tracking / validation.
 *
 * liso_points_in_boxes_f32 replaces
 *   - get_points_in_boxes_mask            (liso/datasets/torch_dataset_commons.py:1902-1935; tracking.py:794-803 sums it
 *                                          per box for the `min_points_in_box` filter),
 *   - Shape.get_points_in_box_bool_mask   (liso/kabsch/shape_utils.py:488-538),
 *   - the mean-flow-per-box reduction of propagate_boxes_forward_using_flow (liso/tracker/tracking.py:2176-2185).
 * The reference forms inv(sensor_T_box) [K,4,4], the box-frame coordinates of every point for every box [N,K,4] and, for
 * the flow mean, a [B,N,K,3] product (144 MB at 120k points x 100 boxes) before reducing over the points.  Here one pass
 * over the points tests each against the boxes held in LDS and accumulates, per box, the number of points inside and the
 * sum of their flow vectors; the [N,K] mask is written only when asked for.
 *
 *   boxes        float32 [B,K,7]  x, y, z, dx, dy, dz, yaw (the dense layout of include/liso_iou3d.h); rows holding NaN
 *                                 contain no point
 *   points       float32 [B,n,point_stride]  x, y, z first; rows holding NaN/inf lie in no box
 *   point_valid  uint8   [B,n] or NULL       only gates the FLOW sum (tracking.py:2181); the point count and the mask
 *                                            ignore it, exactly as the reference's denominator does (:2185)
 *   flow         float32 [B,n,3] or NULL
 *   mask         uint8   [B,n,K] or NULL     1 = point inside box
 *   count        int32   [B,K]   or NULL     points inside (all rows, valid or not)
 *   mean_flow    float32 [B,K,3] or NULL     sum of valid in-box flow / max(count, 1); needs flow, count and workspace
 *   workspace    B*K*3*8 bytes (only for mean_flow): per-box flow sums as 2^-24 m fixed point in int64 -- integer atomics,
 *                so the result does not depend on the order in which blocks finish (bit-reproducible run to run)
 *
 * Inside test: p_box = inv(sensor_T_box) p, inside <=> |p_box| < 0.5 * dims_bloat * dims on x, y, z (strict).  The pose is
 * the yaw-only transform of Shape.get_poses (shape_utils.py:271-319), its inverse is formed in fp64 (closed form);
 * `precision` selects the arithmetic of the matrix-vector product and so which reference function is reproduced:
 *   0: fp64 product, result rounded to fp32 before the comparison   (get_points_in_boxes_mask, :1914-1921)
 *   1: inverse rounded to fp32, fp32 product                         (Shape.get_points_in_box_bool_mask, :514-523)
 */
#ifndef LISO_TRACKING_H
#define LISO_TRACKING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int batch;
    long n;            /* points per batch row */
    int k;             /* boxes per batch row */
    int point_stride;  /* floats per point row, >= 3 */
    int precision;     /* 0 / 1, see above */
    float dims_bloat;  /* box_dims_bloat_factor (1.0 = none) */
} liso_boxpts_cfg;

size_t liso_points_in_boxes_workspace_bytes(const liso_boxpts_cfg* cfg);

int liso_points_in_boxes_f32(const liso_boxpts_cfg* cfg, const float* boxes, const float* points, const uint8_t* point_valid,
                             const float* flow, uint8_t* mask, int* count, float* mean_flow, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---- greedy detection <-> ground-truth matching (validation) -------------------------------------------------------
 * Replaces the "greedy" branch of match_boxes_by_descending_confidence_iou
 * (liso/kabsch/box_groundtruth_matching_iou.py:33-68): predictions are visited in descending confidence; each takes the
 * not-yet-taken ground-truth box of largest IoU (first one on ties) and the pair is a match when that IoU is
 * > threshold.  The reference walks this in Python with an O(n_gt) list lookup per cell; here one wavefront walks it on the
 * device (the IoU matrix, from liso_iou3d_iou_bev_f32, never leaves HBM).
 *
 *   iou          float32, element (g, p) at iou[g * gt_stride + p * pred_stride] (NaN entries are never chosen).  The walk reads
 *                one prediction's column at a time: the [n_pred, n_gt] layout (gt_stride 1) makes those reads contiguous
 *   pred_order   int64 [n_pred]   prediction indices, most confident first
 *   idx_gt, idx_pred int64 [min(n_gt,n_pred)], match_iou float32 [min(n_gt,n_pred)]: the matches in the order found
 *   num_matches  int32 [1]
 *   matched_pred_mask uint8 [n_pred], detected_gt_mask uint8 [n_gt]   (written in full, 0/1)
 */
int liso_match_greedy_f32(const float* iou, long gt_stride, long pred_stride, int n_gt, int n_pred,
                          const int64_t* pred_order, float threshold,
                          int64_t* idx_gt, int64_t* idx_pred, float* match_iou, int* num_matches,
                          uint8_t* matched_pred_mask, uint8_t* detected_gt_mask, void* stream);

/* ---- local box refinement: rectangle fit to the points inside each box ------------------------------------------------
 * Replaces, per box, the body of the loop in perform_local_box_refinement (liso/tracker/tracking.py:2037-2066): points of the
 * sweep whose box-frame x, y (fp64 transform, compared as fp32) lie inside 0.5 * dims_bloat * dims[:2] are handed to
 * fit_2d_box_modest(..., "closeness_to_edge") (liso/box_fitting/box_fitting.py:93-141,242-258): of the 19 headings 0, 5, ..., 90 deg
 * the one whose bounding rectangle has the points closest to its edges (sum of 1 / max(distance, 0.01); first maximum), turned by
 * 90 deg when its y side is the longer one.
 *   points   float32 [n, point_stride] (x, y first); point_valid uint8 [n] or null
 *   boxes    float32 [k, 7] (x, y, z, dx, dy, dz, yaw)
 *   count    int32 [k]    points inside the bloated footprint
 *   fit      float64 [k, 5] = centre x, centre y, length, width, yaw of the fitted rectangle (NaN when count == 0)
 *   workspace  liso_fit_boxes_closeness_workspace_bytes(n, k) bytes, 16-byte aligned (the per-box point lists)
 */
size_t liso_fit_boxes_closeness_workspace_bytes(long n, int k);
int liso_fit_boxes_closeness_f32(const float* points, long n, int point_stride, const uint8_t* point_valid, const float* boxes, int k,
                                 float dims_bloat, int* count, double* fit, void* workspace, size_t workspace_bytes, void* stream);

/* ---- minimum-jerk track smoothing -----------------------------------------------------------------------------------------
 * Replaces the optimisation loop of smooth_track_jerk (liso/tracker/track_smoothing.py:104-230: BatchedSmoothTrack parameters,
 * get_pos_jerk_magnitude, per_batch_mean_loss, torch.optim.Adam(lr) for max_iters steps): all iterations in one launch, one
 * block per track.  observed_pos / smooth_pos float32 [batch, timesteps, 3], valid uint8 [batch, timesteps] (padded frames are free
 * parameters, as in the reference; frame 0 is fixed), timesteps <= 1024.  The loss is the mean over tracks of
 * sum_t valid * |third difference| / n_valid + pos_regul_loss_weight * sum_t valid * |p - observed|^2 / n_valid.
 */
int liso_smooth_tracks_jerk_f32(const float* observed_pos, const uint8_t* valid, int batch, int timesteps, int max_iters,
                                float learning_rate, float pos_regul_loss_weight, float* smooth_pos, void* stream);

/* ---- bicycle-model track smoothing: the rollout and its adjoint ----------------------------------------------------------------
 * Replaces forward_compiled / car_dynamics (liso/tracker/track_smoothing.py:300-337,490-528), the scripted loop inside every loss
 * evaluation of smooth_track_bike_model's L-BFGS (:577-741), and its autograd backward.  One lane per track:
 *   state = (x, y, heading, velocity, heading rate); per frame t < T-1
 *   rate' = clamp(rate + steering[t] dt; -max_yaw_rate, max_yaw_rate), heading' = heading + dt |velocity| / length * rate',
 *   velocity' = clamp(velocity + accel[t] dt; 0, max_velocity), x' = x + velocity' cos(heading') dt, y' likewise with sin;
 *   clamp(u; a, b) = a + (b - a) (0.5 + atan(u / 100) / pi)   (soft_sigmoid_clamp, :28-35).
 * initial_state [batch,5]; accel, steering [batch,T] (column T-1 is not read); vehicle_length [batch]; states [batch,T,5].
 * Backward: grad_states [batch,T,5] -> grad_initial_state [batch,5], grad_accel / grad_steering [batch,T] (column T-1 = 0). */
int liso_bike_rollout_fwd_f32(int batch, int timesteps, const float* initial_state, const float* accel, const float* steering,
                              const float* vehicle_length, float dt, float max_yaw_rate, float max_velocity, float* states, void* stream);
int liso_bike_rollout_bwd_f32(int batch, int timesteps, const float* accel, const float* steering, const float* vehicle_length, float dt,
                              float max_yaw_rate, float max_velocity, const float* states, const float* grad_states,
                              float* grad_initial_state, float* grad_accel, float* grad_steering, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_TRACKING_H */
