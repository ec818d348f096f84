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

/* inv(odom_ta_tb) - I of `batch` row-major 4x4 fp64 matrices (bev_flow_utils.py:30-33, `torch.linalg.inv(odom) - eye`), by cofactor
 * expansion in fp64: the `odom_minus_eye` input of liso_bev_dynamic_flow_f32 without a library LU call (graph-capturable, no host
 * check).  odom, out: [batch,4,4] fp64. */
int liso_odom_inverse_minus_eye_f64(const double* odom, int batch, double* out, void* stream);

/* scratch for liso_fit_box_z_f32 */
size_t liso_fit_box_z_workspace_bytes(int n_points, int n_boxes);

/* fit_bev_box_z_and_height_using_points_in_box for ONE sample: points [N,point_stride] fp32 (all finite),
 * box_pos [K,pos_dims] (2 or 3), box_dims [K,dims_dims] (2 -> height `box_height`, or 3), box_rot [K] ->
 *   num_pts int64 [K], fitted_z fp32 [K], fitted_height fp32 [K] (clipped to [1,2]). */
int liso_fit_box_z_f32(const float* points, int point_stride, int n, const float* box_pos, int pos_dims,
                       const float* box_dims, int dims_dims, const float* box_rot, int k, float box_height,
                       int64_t* num_pts, float* fitted_z, float* fitted_height, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- clustering block (flow_cluster_detector.py:151-189) --------------------------------------------------------------
 * sklearn.cluster.DBSCAN(eps=1.0, min_samples=5, metric="euclidean") on the dynamic pillars' 5-D features
 * (x, y, w*fx, w*fy, w*fz), w = flow_similarity_importance = 2, then skimage.measure.regionprops of the label image
 * (.centroid, .orientation, .axis_major_length, .axis_minor_length) -- both third-party, both on the host in the
 * reference (pins: scikit-learn 0.24.2, scikit-image 0.19.2).  Here the dense BEV grid is the neighbour structure:
 * eps-neighbours of a pillar lie inside a (2*window+1)^2 window, labels are produced on the device with sklearn's
 * numbering (components of the core graph ordered by their first member in row-major order; border pillars take the
 * smallest adjacent cluster), and the region moments are exact integer sums.
 *
 *   dynamic_mask uint8 [B,gx,gy]; row_coords float32 [gx], col_coords float32 [gy] (metric pillar centres: x of a row,
 *   y of a column -- the float32 values of `pcl_bev_center_coords_homog`); flow float32 [B,gx,gy,3] (bev_nonrigid_flow)
 */
typedef struct {
    int batch, gx, gy;
    int window;        /* >= ceil(eps / pillar size): pillars further apart than this in a row or column cannot be neighbours */
    int min_samples;   /* 5 */
    float eps;         /* 1.0 */
    float flow_weight; /* 2.0 (flow_cluster_detector.py:154-160) */
} liso_dbscan_cfg;

/* step 1: core flags uint8 [B,gx,gy], union-find parents int32 [B,gx,gy] (per-sample cell index, -1 for non-core) and
 * is_root int32 [B,gx,gy] (1 where a core pillar is the smallest member of its component). */
int liso_dbscan_components(const liso_dbscan_cfg* cfg, const uint8_t* dynamic_mask, const float* row_coords,
                           const float* col_coords, const float* flow, uint8_t* core, int32_t* parent, int32_t* is_root,
                           void* stream);

/* step 2: root_rank int32 [B,gx,gy] = per-sample inclusive scan of is_root (the caller's scan; 1-based label of a root)
 * -> labels int32 [B,gx,gy]: 0 = background / noise, k >= 1 = sklearn label k-1 (flow_cluster_detector.py:169-172). */
int liso_dbscan_labels(const liso_dbscan_cfg* cfg, const uint8_t* dynamic_mask, const float* row_coords, const float* col_coords,
                       const float* flow, const uint8_t* core, const int32_t* parent, const int32_t* root_rank,
                       int32_t* labels, void* stream);

/* regionprops of a label image: moments uint64 [B,max_labels,6] scratch (n, sum r, sum c, sum r^2, sum c^2, sum rc) ->
 * props float64 [B,max_labels,5] = (centroid_row, centroid_col, orientation, axis_major_length, axis_minor_length);
 * labels above max_labels are ignored, absent labels give zeros. */
int liso_region_props(const int32_t* labels, int batch, int gx, int gy, int max_labels, uint64_t* moments, double* props,
                      void* stream);

/* The same results (bit for bit: the moments are integers) without global atomics: per-block sums in LDS, written to
 * workspace[batch][blocks][max_labels][6] and added per region by a second launch.  For label maps whose labelled cells are scattered
 * over the grid (round 5: 4 306 labelled pillars of an untrained flow network's output in 30 clusters -> 75 us through the atomics of
 * liso_region_props, 26 k of them on 180 addresses; here ~10 us).  0 bytes from the query (more than 1024 labels): the call forwards to
 * liso_region_props. */
size_t liso_region_props_workspace_bytes(int batch, int gx, int gy, int max_labels);
int liso_region_props_ws(const int32_t* labels, int batch, int gx, int gy, int max_labels, uint64_t* moments, double* props,
                         void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_FLOW_CLUSTER_H */
