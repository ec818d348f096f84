/*
 * liso_box_mining.h -- C ABI of the per-box steps between the clustering kernels and the detector targets: what the
 * reference does with ~200 small torch / numpy operations per sweep pair, as a handful of one-block kernels that a captured
 * hipGraph can hold (no scan / sort library call, no memset node, no host read).
 *
 * Replaces, in liso/networks/flow_cluster_detector/flow_cluster_detector.py:
 *   :160-172   np.unique / label bookkeeping after DBSCAN  -> liso_scan_inclusive_i32 (rank of the cluster roots)
 *   :176-206   one box per region: centroid -> pillar centre, axis lengths -> metres, orientation
 *   :208-248   plausibility filters (points, aspect ratio, length, footprint, volume), dropping rejected boxes, padding
 *   :312-331   heading / speed from the per-box Kabsch transform: b0_dT_b1 = inv(T_box) inv(T_bg) T_fg T_box
 *              (liso/kabsch/shape_utils.py:563-605, liso/utils/torch_transformation.py:5-62)
 * and in liso/utils/nms_iou.py:23-66,257-282: ordering by confidence (stable), the pre-NMS cut, and the post-NMS selection
 * around the rotated NMS of include/liso_iou3d.h.
 *
 * Box slots: arrays are [B, K, ...] with K fixed by the caller; `valid` marks the used slots.  Dtypes follow the reference's
 * tensors (pos fp32, dims / rot / probs / velo fp64, class_id / difficulty int32).  Device pointers only; nothing allocates
 * or synchronises; every call enqueues on `stream` and returns LISO_OK or a negative LISO_E* code (include/liso_iou3d.h).
 */
#ifndef LISO_BOX_MINING_H
#define LISO_BOX_MINING_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* inclusive prefix sum along the rows of in[batch][n] (int32) -> out[batch][n]; three launches, fixed order.
 * workspace: liso_scan_workspace_bytes(batch, n). */
size_t liso_scan_workspace_bytes(int batch, long n);
int liso_scan_inclusive_i32(const int32_t* in, int batch, long n, int32_t* out, void* workspace, size_t workspace_bytes,
                            void* stream);

/* One box per labelled region (flow_cluster_detector.py:176-206).
 *   props fp64 [B,K,5] = (centroid_row, centroid_col, orientation, axis_major, axis_minor) of labels 1..K (liso_region_props);
 *   row_coords fp32 [gx] / col_coords fp32 [gy]: metric pillar centres; ppm_x / ppm_y: pillars per metre (fp32 values)
 *   -> center fp32 [B,K,2]: the centre of the pillar that holds the (truncated, clipped) centroid;
 *      dims fp64 [B,K,2] = axis lengths / pillars-per-metre; rot fp64 [B,K]; and their fp32 casts dims_f32 / rot_f32
 *      (what the z-fit of include/liso_flow_cluster.h takes). */
int liso_mine_boxes_from_regions(const double* props, int batch, int k, const float* row_coords, int gx, const float* col_coords,
                                 int gy, float ppm_x, float ppm_y, float* center, double* dims, double* rot, float* dims_f32,
                                 float* rot_f32, void* stream);

typedef struct {
    int batch, k;
    int min_points;            /* >= : flow_cluster_detector.py:209 */
    double aspect_ratio_max;   /* <= : :211-215 */
    double max_box_len_m;      /* <= : :216 */
    double min_box_area_m2;    /* >  : :217 */
    double min_box_volume_m3;  /* >  : :224 */
    int park_invalid;          /* 1: the Kabsch copies of the unused slots are placed 1e6 m away (fixed-slot callers) */
} liso_mine_filter_cfg;

/* Plausibility filters + stable compaction (survivors first, label order kept) + the padded box arrays (:208-248, :311).
 *   num_labels int64 [B]; center fp32 [B,K,2], dims2 fp64 [B,K,2], rot fp64 [B,K] (liso_mine_boxes_from_regions);
 *   num_pts int64 [B,K], fit_z fp32 [B,K], fit_h fp32 [B,K] (liso_fit_box_z_f32)
 *   -> pos fp32 [B,K,3], dims fp64 [B,K,3], rot fp64 [B,K,1], probs fp64 [B,K,1] (1 for survivors), velo fp64 [B,K,1] (0),
 *      valid uint8 [B,K], class_id int32 [B,K,1], difficulty int32 [B,K,1] (unknown-class / invalid ids of shape_utils.py:15-16),
 *      counts int32 [B]; unused slots hold zeros.  kabsch_pos fp32 [B,K,3], kabsch_dims fp32 [B,K,3], kabsch_rot fp32 [B,K]:
 *      fp32 copies for liso_kabsch_trafos_f32 (unused slots parked when cfg->park_invalid). */
int liso_mine_filter_compact(const liso_mine_filter_cfg* cfg, const int64_t* num_labels, const float* center, const double* dims2,
                             const double* rot_in, const int64_t* num_pts, const float* fit_z, const float* fit_h, float* pos,
                             double* dims, double* rot, double* probs, double* velo, uint8_t* valid, int32_t* class_id,
                             int32_t* difficulty, int32_t* counts, float* kabsch_pos, float* kabsch_dims, float* kabsch_rot,
                             void* stream);

/* Heading and speed of every box from its Kabsch transform (:312-331):
 *   trafos fp64 [B, S+1, 4, 4]: slots 0..S-1 the per-box transforms, slot S the background transform (liso_kabsch_trafos_f32);
 *   pos fp32 [B,S,3]; rot fp64 [B,S,1] in/out: rot += atan2(t_y, t_x); velo fp64 [B,S,1] out = |t|, where t is the translation of
 *   inv(T_box) inv(T_bg) (T_fg T_box), T_box = translation(pos) * rotation_z(rot), all fp64. */
int liso_mine_box_motion(const double* trafos, int batch, int s, const float* pos, double* rot, double* velo, void* stream);

/* Confidence order + pre-NMS cut (nms_iou.py:257-266; perform_nms_on_shapes :23-66).  Per sample: slots sorted by descending
 * probs (invalid slots last, stable), every field permuted accordingly IN PLACE (scratch: liso_mine_nms_workspace_bytes), the
 * first `pre_nms_max` (<= 0: all) valid slots enter the NMS:
 *   -> dense fp32 [B,K,7] = (x,y,z,dx,dy,dz,heading) for liso_iou3d_nms_f32 (slots that do not take part are placed far away,
 *      tiny and disjoint: they overlap nothing), enters uint8 [B,K]. */
size_t liso_mine_nms_workspace_bytes(int batch, int k);
int liso_mine_nms_prepare(int batch, int k, int pre_nms_max, float* pos, double* dims, double* rot, double* probs, double* velo,
                          uint8_t* valid, int32_t* class_id, int32_t* difficulty, float* dense, uint8_t* enters, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Post-NMS selection for ONE sample b (nms_iou.py:267-282): keep int64 [K] + num int32 [1] = the survivors of
 * liso_iou3d_nms_f32 on dense[b]; the first `max_boxes` survivors stay valid, every other slot is reset to the padding values
 * (0 / invalid ids, Shape.set_padding_val_to shape_utils.py:439-462).  Also writes the fp32 box arrays the target renderer of
 * include/liso_detector.h takes: t_pos [B,K,3], t_dims [B,K,3] (clamped to >= 1e-3), t_rot [B,K], t_valid uint8 [B,K]. */
int liso_mine_nms_finish(int b, int k, int max_boxes, const int64_t* keep, const int32_t* num, const uint8_t* enters, float* pos,
                         double* dims, double* rot, double* probs, double* velo, uint8_t* valid, int32_t* class_id,
                         int32_t* difficulty, float* t_pos, float* t_dims, float* t_rot, uint8_t* t_valid, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_BOX_MINING_H */
