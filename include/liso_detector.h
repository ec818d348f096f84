/*
 * liso_detector.h -- C ABI of the fused CenterPoint decode + loss for gfx950.
 *
 * Replaces, for the CenterPoint-pillar overlay of the reference (liso/config/liso_config.yml:617-631,693-705:
 * pos = tanh / local_relative_offset, dims = softplus / predict_abs_size, rot = raw 2-vector, probs = raw logit):
 *   activations + decode      liso/networks/simple_net/simple_net.py:111-151, simple_net_utils.py:8-14,
 *                             liso/kabsch/output_modification.py:4-45,58-127
 *   centerpoint_loss          liso/losses/centerpoint_loss.py:13-136   (CenterNet focal :165-200)
 *   rotation_vec_on_unit_circle  liso/kabsch/main_utils.py:51-58
 * which PyTorch runs as ~90 elementwise / reduction launches forward and ~50 backward over [B,128,128,<=3] maps, with
 * boolean-mask indexing (device->host syncs) in the reference.  Here: one pass over the pixels with a fixed-order
 * two-stage reduction forward, one elementwise pass backward.
 *
 * Network maps (pos[3], dims[3], rot[2], probs[1]) are fp32 with arbitrary strides: stride arrays hold, per map,
 * (batch, channel, row, column) strides in floats, so NCHW convolution outputs and NHWC tensors are both read in place.
 * Targets are contiguous [B,H,W,C] fp32 (gt_rot = (sin, cos)); masks are uint8 [B,H,W]; rot_weights [B,H,W] (NULL = 1);
 * pillar_centers [H,W,2] metric centre of every output cell.
 */
#ifndef LISO_DETECTOR_H
#define LISO_DETECTOR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int batch, h, w;
    float res_x, res_y;   /* metres per output cell (bev_range_m / grid) */
    float z_min, z_max;   /* position_representation.box_z_pos_prior_min / _max */
    float sup_weight;     /* loss.supervised.supervised_on_clusters.weight */
    float rot_reg_weight; /* rotation_representation.regul_weight (0: no unit-circle regulariser) */
} liso_centerloss_cfg;

#define LISO_CENTERLOSS_NSUM 12  /* raw sums kept for the backward pass */
#define LISO_CENTERLOSS_NOUT 6   /* probs | rot | dims | pos | rot regulariser | total (weights applied) */

size_t liso_centerloss_workspace_bytes(const liso_centerloss_cfg* cfg);

/* strides: long[16] = {pos: b,c,h,w | dims: b,c,h,w | rot: b,c,h,w | probs: b,c,h,w}.
 * sums: float64 [LISO_CENTERLOSS_NSUM] (out), losses: float32 [LISO_CENTERLOSS_NOUT] (out). */
int liso_centerloss_fwd_f32(const liso_centerloss_cfg* cfg, const float* pos, const float* dims, const float* rot,
                            const float* probs, const long* strides, const float* gt_probs, const float* gt_dims,
                            const float* gt_pos, const float* gt_rot, const uint8_t* center_mask, const uint8_t* ignore_mask,
                            const float* rot_weights, const float* pillar_centers, double* sums, float* losses,
                            void* workspace, size_t workspace_bytes, void* stream);

/* gradients of `total` (scaled by *grad_total, a device scalar) w.r.t. the four network maps, written with the same
 * strides as the inputs */
int liso_centerloss_bwd_f32(const liso_centerloss_cfg* cfg, const float* pos, const float* dims, const float* rot,
                            const float* probs, const long* strides, const float* gt_probs, const float* gt_dims,
                            const float* gt_pos, const float* gt_rot, const uint8_t* center_mask, const uint8_t* ignore_mask,
                            const float* rot_weights, const float* pillar_centers, const double* sums,
                            const float* grad_total, float* g_pos, float* g_dims, float* g_rot, float* g_probs, void* stream);

/* ---- CenterPoint target maps from boxes ---------------------------------------------------------
 * draw_heat_regression_maps (liso/datasets/torch_dataset_commons.py:190-339) with the gaussians of
 * batched_render_gaussian_kabsch_mask (liso/kabsch/kabsch_mask.py:56-116): per box a rotated gaussian with variances
 * 0.15 * (length, width), normalised by its own maximum over the grid (clamped at 1e-5); per cell the hottest box wins
 * (ties add up) where its heat exceeds 0.01.  The reference renders this per sample in DataLoader workers with numpy
 * (a [K,H,W] tensor per sample); here one launch finds the per-box maxima and one renders every cell.
 *   box_pos [B,K,3], box_dims [B,K,3], box_rot [B,K] float32, box_valid uint8 [B,K] (padding slots: 0)
 *   -> probs [B,H,W,1], dims [B,H,W,3], pos [B,H,W,3], rot [B,H,W,2] = (sin, cos), center_mask uint8 [B,H,W]
 *   box_max: float32 [B,K] scratch. */
typedef struct {
    int batch, n_boxes, h, w;
    float range_x, range_y; /* bev_range_m */
} liso_targets_cfg;

int liso_render_center_targets_f32(const liso_targets_cfg* cfg, const float* box_pos, const float* box_dims, const float* box_rot,
                                   const uint8_t* box_valid, float* box_max, float* probs, float* dims, float* pos, float* rot,
                                   uint8_t* center_mask, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_DETECTOR_H */
