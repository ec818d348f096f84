/*
 * liso_slim.h -- C ABI of the MI355X-native SLIM (RAFT-on-BEV) ops.
 *
 * liso_corr_lookup_*: replaces CorrBlock (liso/slim/model/raft_code/corr.py:6-56) + bilinear_sampler
 * (raft_code/utils.py:15-28).  The reference materialises the all-pairs volume fmap1^T fmap2 / sqrt(D)
 * ([B*hw, 1, h, w] fp32: 67 MB at 512^2 BEV, 1.07 GB at 1024^2), average-pools it three times and runs four
 * grid_sample calls per RAFT iteration.  Correlation, pooling and bilinear sampling are all linear in fmap2, so
 *     lookup[p, level i, (a,b)] = < fmap1[p] , bilerp(avgpool^i(fmap2), coords[p]/2^i + (a-r, b-r)) > / sqrt(D)
 * and the volume is never formed: the kernel reads the (tiny, L2-resident) pooled feature maps on the fly.
 *
 * Layouts (all fp32, device pointers):
 *   fmap1      [B, h*w, D]          channels-last query features, D % 4 == 0, D <= 256
 *   fmap2_lvl  [B, H_i, W_i, D]     channels-last pooled target features, H_i = floor(h / 2^i)
 *   coords     [B, 2, h, w]         (x, y) in level-0 pixels (raft_mod.py: pixel_coords_t1)
 *   out        [B, h, w, L*(2r+1)^2] channels-last; channel = i*(2r+1)^2 + a*(2r+1) + b with x offset (a-r) and
 *                                    y offset (b-r)  (the reference's meshgrid(dy, dx) order, corr.py:31-35)
 */
#ifndef LISO_SLIM_H
#define LISO_SLIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISO_CORR_MAX_LEVELS 4

typedef struct {
    int batch, h, w, dim; /* query grid and feature dimension */
    int levels, radius;   /* corr_cfg.num_levels (4), corr_cfg.search_radius (3): window (2r+1)^2, r <= 3 */
} liso_corr_cfg;

int liso_corr_lookup_fwd_f32(const liso_corr_cfg* cfg, const float* fmap1, const float* const* fmap2_levels,
                             const float* coords, float* out, void* stream);

/* The same lookup with the rows of fmap2 shared by 4 x 8 neighbouring query pixels: one block stages the region their windows cover
 * (clipped to the map, <= 256 rows; blocks whose queries lie further apart take the per-query path above) once and multiplies it with
 * the block's 32 query rows on the matrix cores, fp32 operands as bf16 hi / lo pairs (hi hi + hi lo + lo hi: the arithmetic of the
 * F32X3 convolutions of include/liso_conv.h; relative error ~2^-16 of |f1| |f2| per product, against the fp32 FMAs of
 * liso_corr_lookup_fwd_f32).  fmap1 and every level 16-byte aligned.  Same arguments, same output layout. */
int liso_corr_lookup_fwd_tiled_f32(const liso_corr_cfg* cfg, const float* fmap1, const float* const* fmap2_levels,
                                   const float* coords, float* out, void* stream);

/* The pooled target pyramid fmap2_levels[1 .. levels-1] from fmap2 = level 0 ([B, h, w, D] channels last): level i =
 * avg_pool2d(level i-1, 2, stride 2) with floor semantics (H_i = H_{i-1} / 2, an odd last row / column dropped).  The reference pools
 * the correlation volume (corr.py:20-21); pooling is linear in fmap2, so the lookups are the same numbers.  One launch for all levels;
 * levels[0] is not written (pass anything).  All pointers 16-byte aligned. */
int liso_corr_pyramid_fwd_f32(const liso_corr_cfg* cfg, const float* fmap2, float* const* levels, void* stream);
/* Its adjoint in one launch: grad_fmap2 = g_0 + up(g_1 + up(g_2 + ...) / 4) / 4 with g_i = grad_levels[i] ([B, H_i, W_i, D]; NULL =
 * zeros) and up() the replication onto the 2 x 2 pixels each pooled pixel averaged.  grad_fmap2 [B, h, w, D] is overwritten. */
int liso_corr_pyramid_bwd_f32(const liso_corr_cfg* cfg, const float* const* grad_levels, float* grad_fmap2, void* stream);

/* Backward, step 1 of 2.  grad_out has the layout of `out`.  dvol_levels[i] is a dense fp32 matrix
 * [B, h*w, H_i*W_i] (zero-filled by the caller before the first call): the gradient with respect to the pooled
 * correlation volume of level i, into which this call ADDS the adjoint of its bilinear 7x7 windows (<= 64 entries per
 * row and level).  Every entry is owned by one lane, so the update is a plain read-modify-write: no atomics, bit
 * reproducible; calls for the RAFT iterations of one direction accumulate into the same matrices (they share fmap1 and
 * fmap2).  Step 2, once per direction: liso_corr_bwd_features_f32 below.  coords receive no gradient (detached per iteration,
 * raft_mod.py:189). */
int liso_corr_lookup_bwd_dvol_f32(const liso_corr_cfg* cfg, const float* coords, const float* grad_out,
                                  float* const* dvol_levels, void* stream);

/* Backward, step 2 of 2: the adjoint of corr = fmap1^T fmap2 / sqrt(D) (liso/slim/model/raft_code/corr.py:48-56; pooled per level,
 * :20-21 -- the 1 / sqrt(D) is already inside dvol), once per flow direction:
 *     grad_fmap1[b]         [h*w, D]     = sum_i  dvol_i[b] [h*w, H_i*W_i]   . fmap2_levels[i][b] [H_i*W_i, D]
 *     grad_fmap2_levels[i][b] [H_i*W_i, D] =      dvol_i[b]^T [H_i*W_i, h*w] . fmap1[b] [h*w, D]
 * Two launches of one 128 x 128-tile matrix-core kernel (all levels in each; K split to fill the chip, partial tiles added in a fixed
 * order by a second kernel: no float atomics, bitwise reproducible).  `mode`: LISO_CONV_F32X3 (operands split into bf16 hi / lo,
 * three MFMAs per product: the arithmetic of the convolutions of include/liso_conv.h) or LISO_CONV_F32 (exact fp32 MFMA).  fmap1,
 * the levels and the outputs 16-byte aligned; any h, w (levels whose H_i*W_i is not a multiple of 4 take scalar loads).  Outputs are
 * overwritten.  workspace: liso_corr_bwd_features_workspace_bytes(cfg) bytes. */
size_t liso_corr_bwd_features_workspace_bytes(const liso_corr_cfg* cfg);
int liso_corr_bwd_features_f32(const liso_corr_cfg* cfg, int mode, const float* fmap1, const float* const* fmap2_levels,
                               const float* const* dvol_levels, float* grad_fmap1, float* const* grad_fmap2_levels, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- exact 1-nearest-neighbour search (SLIM self-supervised loss) -------------------------------------------------
 * Replaces knn_graph(x, index=ref, k=1, loop=True) (liso/slim/slim_loss/knn_graph.py:10-98), which copies both clouds
 * to the host and queries a pynanoflann KD-tree per call (knn_wrapper.py:139-152,180-186): 12+ device->host->device
 * round trips per training step.  Here the reference cloud is bucketed once into a uniform xy grid (with z bins inside
 * every cell) on the device and every query walks Chebyshev rings of cells, reading only the z bins that can still
 * hold a closer point, until the best squared distance provably cannot be improved: EXACT
 * nearest neighbour (3-D Euclidean), ties resolved towards the smaller reference index, no host involvement.
 *
 *   ref    float32 [n_ref, ref_stride]   (x,y,z first); rows containing NaN/inf (padding) are never returned
 *   query  float32 [n_query, query_stride], rows containing NaN/inf get index 0 and distance NaN
 *   index  int64 [n_query], dist_sqr float32 [n_query] (may be NULL)
 */
typedef struct {
    float x_min, y_min;  /* grid origin */
    float cell;          /* xy cell edge length in metres (> 0) */
    int nx, ny;          /* xy cells; reference points outside are clamped into the border cells (still exact) */
    float z_min, z_cell; /* z bins inside every xy cell: bin = clamp(floor((z - z_min) / z_cell), 0, nz-1) */
    int nz;              /* >= 1; nx*ny*nz <= 2^24.  Bins only prune candidates: any nz gives the exact answer */
} liso_knn_grid;

size_t liso_knn_workspace_bytes(const liso_knn_grid* grid, int n_ref);

/* bucket the reference cloud (counting sort by cell); the workspace then holds the index structure */
int liso_knn_build_f32(const liso_knn_grid* grid, const float* ref, int ref_stride, int n_ref, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The reference rows in bucket order (cell-major, z bin inside the cell): ids[0 .. n_ref) int64.  Only meaningful when every
 * reference row was finite (rows with NaN/inf are not bucketed: the tail of `ids` is then undefined).  Use: a spatially
 * coherent processing order for queries that start at this cloud's points (liso_amd/slim/slim_loss/knn_wrapper.py). */
int liso_knn_sorted_ids(const liso_knn_grid* grid, const void* workspace, int n_ref, int64_t* ids, void* stream);

/* One launch answers all queries.  `grid`/`workspace`: the (fine) index.  `coarse_grid`/`coarse_workspace` (both NULL or
 * both set): a second index of the SAME reference cloud with larger cells -- a query whose answer is not proven exact
 * after `fine_max_rings` rings of the fine grid continues on the coarse one (the fine grid answers the dense near field
 * in 1-2 rings, the coarse grid the few queries that land in empty space, so no query walks thousands of empty cells).
 * Without a coarse index the fine search runs until proven exact (fine_max_rings ignored). */
int liso_knn_query_f32(const liso_knn_grid* grid, const void* workspace, const liso_knn_grid* coarse_grid,
                       const void* coarse_workspace, int n_ref, const float* query, int query_stride, int n_query,
                       int64_t* index, float* dist_sqr, int fine_max_rings, void* stream);

/* ---- nearest-point flow loss ---------------------------------------------------------------------------------------
 * compute_flow_loss_a_to_b + NearestPointLoss + huber_delta (liso/slim/slim_loss/knn_wrapper.py:11-49,58-135,155-217) after
 * the 1-NN indices are known: q = cloud_a + flow, nn = cloud_b[b, index], d2 = |nn - q|^2, field-of-view weight from the
 * BEV extent, loss = huber(d2) * weight.  ~30 torch launches forward and ~40 backward per call become one each.
 *   cloud_a, flow float32 [B,n,3] (padding rows NaN -> loss/d2 NaN, gradient 0); cloud_b float32 [B,n_b,3];
 *   index int64 [B,n]; loss, dist_sqr float32 [B,n].
 * Backward: grad_loss [B,n] (and optionally grad_dist_sqr [B,n], may be NULL) -> grad_flow [B,n,3] (== grad cloud_a). */
typedef struct {
    int batch;
    long n, n_b;
    float ext[4];   /* bev extent x_min, y_min, x_max, y_max */
    int fov_mode;   /* 0 none, 1 ignore_out_fov, 2 mask_close_fov  ("use_nearest" is not covered: torch path) */
    float delta;    /* L1_delta of huber_delta(mode "large_grad_1"); 0 = plain (gradient-safe) norm */
} liso_nploss_cfg;

int liso_nearest_point_loss_fwd_f32(const liso_nploss_cfg* cfg, const float* cloud_a, const float* flow, const float* cloud_b,
                                    const int64_t* index, float* loss, float* dist_sqr, void* stream);
int liso_nearest_point_loss_bwd_f32(const liso_nploss_cfg* cfg, const float* cloud_a, const float* flow, const float* cloud_b,
                                    const int64_t* index, const float* grad_loss, const float* grad_dist_sqr, float* grad_flow,
                                    void* stream);

/* ---- BEV grid -> per-point gather (decoder) and its adjoint ---------------------------------------------------------
 * Replaces batched_grid_data_to_pointwise_data (liso/slim/slim_loss/static_aggregation.py:8-31; used by
 * head_decoder.py:300-408 to pull 26 channels per point out of the decoded BEV maps).  PyTorch's advanced-index backward
 * sorts the 120k indices and runs a serial-per-segment accumulate on every call (3 x 0.3 ms per decode, 12 decodes per
 * step, measured); here the points are sorted by cell once per cloud and every backward is one segmented-sum launch.
 *
 *   grid        float32 [cells_total, c]  channels-last BEV maps of the whole batch flattened (cells_total = B*H*W)
 *   lin         int32 [n_rows]            flattened cell of every point (b*H*W + row*W + col), < 0: invalid point
 *   out         float32 [n_rows, c]       invalid rows receive `default_value`
 *   sorted_lin  int32 [n_rows] ascending  (lin sorted), order int32 [n_rows] = the permutation that sorts lin,
 *   seg_rank    int32 [n_rows]            position of the sorted row inside its run of equal cells (0 = first)
 *   partial     float32 [n_rows, c]       scratch
 *   grad_grid   float32 [cells_total, c]  ZERO-FILLED by the caller; rows of cells holding >= 1 point are overwritten
 */
int liso_bev_gather_fwd_f32(const float* grid, const int* lin, long n_rows, int c, float default_value, float* out,
                            void* stream);
/* `lin` of a batch of clouds: coors int32 / int64 [batch, n, 2] = (row, col) of every point on the h x w grid, valid uint8 / bool
 * [batch, n] -> lin int32 [batch * n] = (b h + row) w + col, -1 for invalid points. */
int liso_bev_lin_index(const void* coors, int coors_are_int64, const unsigned char* valid, int batch, long n, int h, int w, int* lin,
                       void* stream);

/* The gather plan's index arithmetic (BevGatherPlan.tiled / _sort, liso_amd/slim/slim_loss/static_aggregation.py; the reference indexes
 * grid[b, row, col] per point and lets autograd scatter: liso/slim/slim_loss/static_aggregation.py:69-84).
 * tile_lin: lin of the tiled batch [samples[:half]] * n_it + [samples[half:]] * n_it from the distinct samples' rows [n2, n] (cells of copy j
 *   shifted by (j - its sample) * h * w, invalid rows stay -1).
 * rank: rank[i] = i - (first index of sorted_lin[i] in the ascending list); order32 (optional) = order64 narrowed.
 * expand: the cell-sorted view (sorted_lin, order, rank: [n2 * n_it * n] each) of the tiled batch from ONE stable sort of the distinct
 *   samples' flat list (s_flat ascending: invalid rows first, then sample 0's cells, sample 1's, ...; o_flat its permutation; rank_flat
 *   from `rank`): every copy's block holds its sample's valid rows in sorted order, then -1 padding. */
int liso_bev_plan_tile_lin(const int* rows, int n2, long n, int n_it, int half, int h, int w, int* lin, void* stream);
int liso_bev_plan_rank(const int* sorted_lin, const long long* order64, long n, int* rank, int* order32, void* stream);
int liso_bev_plan_expand(const int* s_flat, const long long* o_flat, const int* rank_flat, int n2, long n, int n_it, int half, int h, int w,
                         int* sorted_lin, int* order, int* rank, void* stream);
int liso_bev_gather_bwd_f32(const float* grad_out, const int* sorted_lin, const int* order, const int* seg_rank, long n_rows,
                            int c, float* partial, float* grad_grid, void* stream);

/* ---- RAFT output assembly -------------------------------------------------------------------------------------------
 * Replaces, for all update iterations of a step at once, upflow_n / uplogits_n (liso/slim/model/raft_code/utils.py:5-12:
 * x`factor` bilinear F.interpolate, align_corners=True), change_flow_convention_from_raft2usfl (raft_mod.py:262-266),
 * HeadDecoder.concat2network_output (head_decoder.py:37-65) and the channels-last permute (raft_mod.py:244-258): per
 * iteration the reference runs 2 upsamplings, a flip, 2 scalings and a concat over 512x512 maps (and autograd their six
 * adjoints); here one launch writes the [S, H, W, 8] network outputs of all iterations, two launches form the adjoint.
 *
 *   flow_lr    float32 [n_it, batch2, 2, h, w]   coords1 - coords0 per iteration, RAFT convention (x = col, y = row), pixels
 *   logits_lr  float32 [n_it, batch2, 4, h, w]
 *   out        float32 [n_it*batch2, H, W, 8], H = h*factor: channels 0:4 logits, 4:6 static flow (row, col) in metres,
 *              6:8 dynamic flow (the same values).  Sample order: output sample dir*(n_it*B) + it*B + b holds input sample
 *              (it, dir*B + b), B = batch2/dirs -- with dirs = 2 the forward-flow samples of all iterations come first,
 *              then the backward-flow ones (the order the stacked loss consumes); dirs = 1 keeps (it, b) order.
 *   flow_scale = factor * metres per low-resolution pixel
 * Bilinear taps follow ATen's upsample_bilinear2d (align_corners=True) arithmetic. */
typedef struct {
    int n_it, batch2, dirs;
    int h, w, factor;
    float flow_scale;
} liso_upsample_cfg;

size_t liso_raft_upsample_scratch_bytes(const liso_upsample_cfg* cfg);
int liso_raft_upsample_outputs_fwd_f32(const liso_upsample_cfg* cfg, const float* flow_lr, const float* logits_lr, float* out,
                                       void* stream);
/* grad_out [S,H,W,8] -> grad_flow_lr [n_it,batch2,2,h,w], grad_logits_lr [n_it,batch2,4,h,w] (overwritten); two gather
 * passes (along x, then y) through `scratch`: no atomics, bit reproducible. */
int liso_raft_upsample_outputs_bwd_f32(const liso_upsample_cfg* cfg, const float* grad_out, void* scratch, size_t scratch_bytes,
                                       float* grad_flow_lr, float* grad_logits_lr, void* stream);

/* ---- ConvGRU gates of the update block ---------------------------------------------------------------------------------
 * Replaces the elementwise part of ConvGRU.forward (liso/slim/model/update.py:29-37) between its three convolutions:
 *     z = sigmoid(convz(hx)); r = sigmoid(convr(hx)); q = tanh(convq(cat([r*h, x]))); h' = (1-z)*h + z*q
 * 10 ATen launches forward and ~15 backward per RAFT iteration become 2 + 2.  All maps NCHW fp32 (hw = H*W):
 *   cz, cr  [B,ch,hw]   pre-activations; `zr_batch_stride` (elements) lets them be the channel halves of ONE merged
 *                       convolution output [B,2ch,hw] (convz and convr read the same input)
 *   h [B,ch,hw], x [B,cx,hw];  z [B,ch,hw] (saved for the output gate);  rhx [B,ch+cx,hw] = cat([r*h, x]) for convq
 * in_bwd: g_rhx [B,ch+cx,hw] (+ g_z [B,ch,hw] or NULL) -> g_cz, g_cr (batch stride `gzr_batch_stride`), g_h [B,ch,hw];
 *         the gradient of x is the tail of g_rhx.   out_fwd/out_bwd: n = B*ch*hw contiguous elements. */
typedef struct {
    int batch, ch, cx;
    long hw;
} liso_gru_cfg;

int liso_gru_in_fwd_f32(const liso_gru_cfg* cfg, const float* cz, const float* cr, long zr_batch_stride, const float* h,
                        const float* x, float* z, float* rhx, void* stream);
int liso_gru_in_bwd_f32(const liso_gru_cfg* cfg, const float* cr, long zr_batch_stride, const float* h, const float* z,
                        const float* g_z, const float* g_rhx, float* g_cz, float* g_cr, long gzr_batch_stride, float* g_h,
                        void* stream);
int liso_gru_out_fwd_f32(long n, const float* cq, const float* z, const float* h, float* out, void* stream);
int liso_gru_out_bwd_f32(long n, const float* cq, const float* z, const float* h, const float* g_out, float* g_cq, float* g_z,
                         float* g_h, void* stream);

/* The same gate arithmetic at inference on PIXEL ROWS that are channel slices of wider channels-last buffers (strides in floats
 * between consecutive pixels): z[n_pix, ch] = sigmoid(zr[:, :ch]), rh = sigmoid(zr[:, ch:2ch]) * h;  h <- (1 - z) h + z tanh(cq) in
 * place.  With them the update block keeps [h | inp | out | class | flow | r*h] in ONE buffer (the convolutions write their channel
 * ranges, liso_conv.h): none of the four concatenations of liso/slim/model/update.py:29-37,84-96,139-141 per iteration. */
int liso_gru_in_rows_f32(long n_pix, int ch, const float* zr, long zr_stride, const float* h, long h_stride, float* z, float* rh,
                         long rh_stride, void* stream);
int liso_gru_out_rows_f32(long n_pix, int ch, const float* cq, long cq_stride, const float* z, float* h, long h_stride, void* stream);

/* The RAFT loop's state update at inference (liso/slim/model/raft.py:199-216: coords1 = coords1 + delta_flow, logits = logits +
 * delta_logits, and the next iteration's flow = coords1 - coords0) as ONE launch, the same fp32 operations in the same order.
 * delta: batch * hw pixels of (flow x, flow y, logit 0..3) at `delta_stride` floats per pixel (the heads' merged output);
 * coords0 / coords1: [batch, 2, hw] (coords1 updated in place: the correlation lookup reads it);
 * state8: [batch * hw][8] = (flow x, flow y, logit 0..3, 0, 0), 16-byte aligned: flow overwritten, logits accumulated -- the
 * channels-last input of the motion encoder's 7x7 convolutions (update.py:54-59 on one map). */
int liso_raft_state_step_f32(int batch, int hw, const float* delta, long delta_stride, const float* coords0, float* coords1,
                             float* state8, void* stream);

/* ---- the RAFT update loop's TRAINING step between its convolutions (liso_amd/slim/model/raft_loop.py: all iterations of
 * liso/slim/model/raft.py:178-259 + update.py:29-164 as one autograd node on stacked channels-last buffers) ------------------------
 * Rows = pixels; every pointer may address a channel slice of a wider buffer (`*_stride` = floats between consecutive pixels,
 * multiples of 4, pointers 16-byte aligned).
 *
 * liso_rows_combine_f32: out[p, :] = (a[p, :] + b[p, :] + c[p, :]) * (mask[p, :] > 0), added to out when `accumulate`; b, c, mask may
 * be NULL.  One launch for what autograd spends a gradient accumulation, a ReLU backward (threshold_backward) and a concatenation
 * on: the masked gradient lands in the stacked buffer the layer's weight gradient reads. */
int liso_rows_combine_f32(long n_pix, int channels, const float* a, long a_stride, const float* b, long b_stride, const float* c,
                          long c_stride, const float* mask, long mask_stride, float* out, long out_stride, int accumulate, void* stream);
/* ConvGRU output gate out of place (update.py:35-37): h_out = (1 - z) h_in + z tanh(cq); z [n_pix, ch] contiguous. */
int liso_gru_out_rows_train_f32(long n_pix, int ch, const float* cq, long cq_stride, const float* z, const float* h_in, long h_in_stride,
                                float* h_out, long h_out_stride, void* stream);
/* its adjoint: g_cq = g z (1 - tanh(cq)^2) (strided), g_z = g (tanh(cq) - h_in), g_h = g (1 - z) (both [n_pix, ch] contiguous) */
int liso_gru_out_rows_bwd_f32(long n_pix, int ch, const float* cq, long cq_stride, const float* z, const float* h_in, long h_in_stride,
                              const float* g_out, long g_out_stride, float* g_cq, long g_cq_stride, float* g_z, float* g_h, void* stream);
/* adjoint of liso_gru_in_rows_f32: zr [.., 2 ch] pre-activations of z | r; g_zr[:, :ch] = g_z z (1 - z), g_zr[:, ch:] = g_rh h r (1 - r),
 * g_h = g_rh r ([n_pix, ch] contiguous) */
int liso_gru_in_rows_bwd_f32(long n_pix, int ch, const float* zr, long zr_stride, const float* h, long h_stride, const float* z,
                             const float* g_z, const float* g_rh, long g_rh_stride, float* g_zr, long g_zr_stride, float* g_h, void* stream);
/* The loop's state update (raft.py:199-216) keeping every iteration's state.  State pixel (8 floats) = (logit 0..3, flow x, flow y, 0, 0);
 * delta8 = the heads' merged output in the same order.  coords_out = coords_in + delta flow ([batch, 2, hw]); state_out = (state_in
 * logits + delta logits, coords_out - coords0, 0, 0); flow_out [batch, 2, hw] / logits_out [batch, 4, hw]: the same numbers planar,
 * what liso_raft_upsample_outputs_fwd_f32 reads. */
int liso_raft_state_step_train_f32(int batch, int hw, const float* delta8, const float* coords0, const float* coords_in, float* coords_out,
                                   const float* state_in, float* state_out, float* flow_out, float* logits_out, void* stream);
/* planar output gradients (g_flow [n, 2, hw], g_logits [n, 4, hw]) -> g8 [n * hw][8] = (logits | flow | 0 0): the gradient of the heads'
 * merged output convolution in its own layout, all iterations in one launch */
int liso_raft_pack_output_grads_f32(long n, int hw, const float* g_flow, const float* g_logits, float* g8, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_SLIM_H */
