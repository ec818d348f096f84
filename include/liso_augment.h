/* Box-snippet augmentation on the device (second half).
 *
 * Replaces the numpy / scikit-image body of LidarDataset.create_augmented_sample_from_box_snippet_db
 * (liso/datasets/torch_dataset_commons.py:1531-1776), which the reference runs per sample in DataLoader workers:
 *   - the "where may an object centre go" mask: BEV occupancy of the sweep, dilated with a disk, inverted  (:1538-1557)
 *   - picking the drawn free cells out of the row-major list of free cells                                 (:1558-1563)
 *   - pasting the drawn snippets: per-point gather from the snippet database, rigid pose x flip x scale in float64,
 *     artificial per-point flow and the box speed = mean |flow|                                             (:1597-1690)
 * The random draws themselves stay with the caller (liso_amd/datasets/box_augmentation.py draws them in the reference's
 * order so seeded runs reproduce the reference; a device generator can feed the same entry points).
 *
 * Plain C ABI: device pointers, sizes, a hipStream_t passed as void*.  Return 0 = LISO_OK, else the codes of liso_iou3d.h.
 */
#ifndef LISO_AUGMENT_H
#define LISO_AUGMENT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of scratch for liso_bev_free_mask (the occupancy byte map) */
size_t liso_bev_free_mask_workspace_bytes(int h, int w);

/* free_mask[i * w + j] = 1 iff no occupied BEV cell (i', j') has (i - i')^2 + (j - j')^2 <= radius^2
 * (= ~skimage.morphology.binary_dilation(occupancy, disk(radius)); cells outside the grid count as empty).
 *   pillar_coors   int32 [n, 2] (row, col) of every point of the sweep; rows with a coordinate outside the grid are ignored
 *   free_mask      uint8 [h * w]
 *   row_free_prefix int32 [h + 1]: exclusive prefix sum of the free cells per row; row_free_prefix[h] = number of free cells
 */
int liso_bev_free_mask(const int32_t* pillar_coors, long n, int h, int w, int radius, uint8_t* free_mask,
                       int32_t* row_free_prefix, void* workspace, size_t workspace_bytes, void* stream);

/* flat_cell[q] = i * w + j of the compact_idx[q]-th free cell in row-major order (what indexing
 * `pcl_bev_center_coords_homog_np[valid_mask][idx]` selects, :1558-1563); -1 when compact_idx[q] is out of range */
int liso_bev_select_free_cells(const uint8_t* free_mask, const int32_t* row_free_prefix, int h, int w,
                               const int64_t* compact_idx, int k, int32_t* flat_cell, void* stream);

/* Paste k snippets.  Object i owns the output rows [out_offsets[i], out_offsets[i + 1]).
 *   db_points    float32 [T, 4]   all snippets of the database, box coordinates + intensity, concatenated
 *   src_index    int64 [n_out]    database row of every output point (the caller's point drop-out / ray-drop selection)
 *   pose         float64 [k, 12]  rows 0..2 of sensor_T_box x diag(flip_x scale_x, flip_y scale_y, scale_z, 1)
 *   flow_rand    float64 [n_out, 3] uniform [0, 1) draws: flow = vmin + rand * (vmax - vmin)                (:1672-1677)
 *   out_points   float32 [n_out, 4] (float64 arithmetic, rounded once, like the reference's .astype(np.float32))
 *   out_flow     float32 [n_out, 3] or null
 *   box_velo     float32 [k]      mean over the object's points of |flow| (:1678-1682), fixed summation order
 */
int liso_snippet_paste(const float* db_points, long db_rows, const int64_t* src_index, const int64_t* out_offsets,
                       const double* pose, const double* flow_rand, double vmin, double vmax, int k, float* out_points,
                       float* out_flow, float* box_velo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_AUGMENT_H */
