/*
 * liso_flow_cluster.h -- C ABI of the MI355X-native flow -> pseudo-box stages around the clustering step.
 *
 * Replaces, for the reference call sites in liso/networks/flow_cluster_detector/flow_cluster_detector.py:87-384:
 *   get_bev_dynamic_flow_map_from_pcl_flow_and_odom   liso/utils/bev_flow_utils.py:6-77
 *   masked_scatter_mean_2d / scatter_add_2d           liso/utils/torch_differentiable_forward_scatter.py:22-87
 *   fit_bev_box_z_and_height_using_points_in_box      flow_cluster_detector.py:339-384
 * Device pointers only; nothing allocates or synchronises; returns LISO_OK or a negative code (liso_iou3d.h).
 */
#ifndef LISO_FLOW_CLUSTER_H
#define LISO_FLOW_CLUSTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of scratch for liso_bev_dynamic_flow_f32: int64 fixed-point sums [B,H,W,4] + int32 counts [B,H,W] */
size_t liso_bev_dynamic_flow_workspace_bytes(int batch, int h, int w);

/* Per point: nonrigid flow = point flow - (inv(odom) - I) p (fp64, bev_flow_utils.py:28-41), then the per-pillar mean
 * of |nonrigid| and of nonrigid (divided only where count > 1, torch_differentiable_forward_scatter.py:84-86).
 *   points [B,N,point_stride] fp32 (NaN padding allowed), valid [B,N] uint8, pillar_coors [B,N,2] int32 (row, col),
 *   flow [B,N,flow_stride>=3] fp32, odom_minus_eye [B,4,4] fp64 = inv(odom_ta_tb) - I (row-major)
 *   -> dynamicness fp32 [B,H,W,1], nonrigid_flow fp32 [B,H,W,3]
 * The scatter uses 64-bit fixed-point integer atomics (2^-24 m resolution): the result does not depend on the order
 * of the atomics, unlike the reference's float scatter_add_. */
int liso_bev_dynamic_flow_f32(const float* points, int point_stride, const uint8_t* valid, const int32_t* pillar_coors,
                              const float* flow, int flow_stride, const double* odom_minus_eye, int batch, int n, int h,
                              int w, float* dynamicness, float* nonrigid_flow, void* workspace, size_t workspace_bytes,
                              void* stream);

/* scratch for liso_fit_box_z_f32 */
size_t liso_fit_box_z_workspace_bytes(int n_points, int n_boxes);

/* fit_bev_box_z_and_height_using_points_in_box for ONE sample: points [N,point_stride] fp32 (all finite),
 * box_pos [K,pos_dims] (2 or 3), box_dims [K,dims_dims] (2 -> height `box_height`, or 3), box_rot [K] ->
 *   num_pts int64 [K], fitted_z fp32 [K], fitted_height fp32 [K] (clipped to [1,2]). */
int liso_fit_box_z_f32(const float* points, int point_stride, int n, const float* box_pos, int pos_dims,
                       const float* box_dims, int dims_dims, const float* box_rot, int k, float box_height,
                       int64_t* num_pts, float* fitted_z, float* fitted_height, void* workspace, size_t workspace_bytes,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_FLOW_CLUSTER_H */
