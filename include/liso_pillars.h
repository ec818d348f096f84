/*
 * liso_pillars.h -- C ABI of the MI355X-native pillar path:
 *   hard voxelisation -> PillarFeatureNet (decorate, Linear, BatchNorm1d, ReLU, max) -> dense BEV scatter,
 * forward and backward, without materialising voxels[P,20,C], [P,20,10] or [P,20,64].
 *
 * Replaces, for the reference call site liso/networks/pcl_to_feature_grid/pcl_to_feature_grid.py:58-107:
 *   mmcv.ops.Voxelization (hard, max_num_points=20, max_voxels=40000; semantics of the in-tree CPU twin
 *     mmdetection3d/mmdet3d/core/voxel/voxel_generator.py:211-280: first-come-first-served in point order)
 *   PillarFeatureNet.forward  mmdetection3d/mmdet3d/models/voxel_encoders/pillar_encoder.py:93-159 (legacy=True)
 *   PFNLayer.forward          mmdetection3d/mmdet3d/models/voxel_encoders/utils.py:146-182
 *   PointPillarsScatter.forward_batch  mmdetection3d/mmdet3d/models/middle_encoders/pillar_scatter.py:62-102
 *
 * Layout
 *   points      float32 [n_total, C]   all samples of the batch concatenated, C in {3,4,5} (x,y,z[,i[,t]])
 *   offsets     host int [B+1]         sample b owns points [offsets[b], offsets[b+1])
 *   voxel rows  fixed stride: sample b owns rows [b*max_voxels, b*max_voxels + num_voxels[b])
 *   coors       int32 [B*max_voxels,4] (b, 0, x_idx, y_idx)  -- the reference's order after its x/y swap (:73,:79-83)
 *   slots       int32 [B*max_voxels, max_points] global point index, ascending; first num_points[v] are valid
 *   canvas      OutT  [B, gx, gy, 64]  channels-last storage of the reference's [B,64,gx,gy] (dim2 = x_idx)
 *   occupancy   float32 [B, gx, gy]
 * All pointers except `offsets` are device pointers.  Nothing allocates or synchronises; every entry point
 * enqueues on `stream` and returns LISO_OK or a negative code (see liso_iou3d.h).
 */
#ifndef LISO_PILLARS_H
#define LISO_PILLARS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISO_PFN_OUT 64        /* PFN output channels (feat_channels=[64], pcl_to_feature_grid.py:41-48) */
#define LISO_PILLARS_MAX_BATCH 32
#define LISO_PFN_STATS_DOUBLES 80 /* >= (C+7)(C+8)/2 second moments of the augmented feature vector */

typedef struct {
    float x_min, y_min, z_min; /* point_cloud_range[0:3] */
    float vx, vy, vz;          /* voxel_size */
    int gx, gy;                /* grid (z grid is 1: voxel z-size spans the whole range) */
    int max_points;            /* 20 */
    int max_voxels;            /* 40000 */
    int n_channels;            /* C */
} liso_pillar_cfg;

/* bytes of device scratch for liso_pillars_voxelize_f32 */
size_t liso_pillars_voxelize_workspace_bytes(const liso_pillar_cfg* cfg, int batch, int n_total);

/* Hard voxelisation (deterministic: voxels ordered by their first point, the first max_points points of a
 * voxel in point order are kept, voxels beyond max_voxels dropped -- voxel_generator.py:249-279).
 * Outputs: coors, num_points [B*max_voxels], slots, num_voxels [B], cell_to_voxel [B*gx*gy] (row+1, 0 = empty). */
int liso_pillars_voxelize_f32(const float* points, const int* offsets_host, int batch, const liso_pillar_cfg* cfg,
                              int* coors, int* num_points, int* slots, int* num_voxels, int* cell_to_voxel,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Decorated feature rows in pillar order (CSR).  Walks the count -> slot indices -> points chain of every pillar once
 * (pillar_encoder.py:109-146: cluster-centre and voxel-centre offsets, legacy aliasing) and writes, for every kept point,
 * a 12-float row [f_0 .. f_{C+5}, 1, 0 pad]; the rows of pillar row v are feat[pt_off[v] .. pt_off[v+1]).
 *   pt_off     int32 [B*max_voxels + 1]   (rows of a sample beyond its num_voxels are empty)
 *   feat       float32 [>= total points, 12]  (16-B aligned)
 *   voxel_cell int32 [B*max_voxels]       flattened canvas cell (b*gx + x_idx)*gy + y_idx of the pillar, -1 if unused
 * Everything downstream (statistics, forward, backward) streams these rows; the point cloud is not read again. */
size_t liso_pfn_decorate_workspace_bytes(int batch, int max_voxels);
int liso_pfn_decorate_f32(const float* points, const liso_pillar_cfg* cfg, int batch, const int* coors, const int* num_points,
                          const int* slots, const int* num_voxels, int* pt_off, float* feat, int* voxel_cell, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Batch statistics of the PFN linear output for BatchNorm1d in training mode, then scale/shift.
 *   weight [64, C+6] row-major, gamma/beta/running_mean/running_var [64] (running_* updated in place when
 *   training != 0 with `momentum`; unbiased variance, as torch.nn.BatchNorm1d).
 *   bn_out float32 [4*64] = scale | shift | mean | invstd ;  moments float64 [LISO_PFN_STATS_DOUBLES] (for backward)
 *   partials: device scratch of liso_pfn_partials_bytes() bytes.
 * training == 0: scale/shift from running stats, nothing else is touched (feat / pt_off may be NULL). */
size_t liso_pfn_partials_bytes(void);
int liso_pfn_bn_prepare_f32(const float* feat, const int* pt_off, const liso_pillar_cfg* cfg, int batch, const int* num_voxels,
                            const float* weight, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, int training, float* bn_out, double* moments,
                            void* partials, void* stream);

/* Fused Linear + BN + ReLU + max + dense scatter.  Every cell of canvas and occupancy is written exactly once (zeros
 * for empty cells: pillar_scatter.py:78-82 allocates zeros), so the caller need NOT pre-fill them.
 * cell_to_voxel comes from liso_pillars_voxelize_f32.  out_bf16 != 0: canvas is bfloat16, else float32. */
int liso_pfn_forward_scatter(const float* feat, const int* pt_off, const int* voxel_cell, const liso_pillar_cfg* cfg, int batch,
                             const int* cell_to_voxel, const float* weight, const float* bn_out, void* canvas, int out_bf16,
                             float* occupancy, void* stream);

/* Backward of the fused op w.r.t. weight, gamma, beta (inputs carry no gradient: voxelize is no_grad,
 * pcl_to_feature_grid.py:56).  grad_canvas has the canvas layout/dtype.  grad_weight [64, C+6],
 * grad_gamma/grad_beta [64] are overwritten.  partials: liso_pfn_partials_bytes() bytes of scratch. */
int liso_pfn_backward(const float* feat, const int* pt_off, const int* voxel_cell, const liso_pillar_cfg* cfg, int batch,
                      const int* num_voxels, const float* weight, const float* gamma, const float* bn_out,
                      const double* moments, int training, const void* grad_canvas, int grad_bf16, float* grad_weight,
                      float* grad_gamma, float* grad_beta, void* partials, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_PILLARS_H */
