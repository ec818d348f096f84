/*
 * liso_kabsch.h -- C ABI of the MI355X-native Kabsch / weighted point-cloud alignment ops.
 *
 * Replaces, for the reference call sites
 *   liso/kabsch/kabsch_mask.py:149-228   render_soft_kabsch_mask_torch / get_box_pixel_weights  (soft box masks)
 *   liso/kabsch/kabsch_mask.py:328-399   KabschDecoder.get_kabsch_trafos_from_point_flow        (fg/bg weights)
 *   liso/kabsch/kabsch_mask.py:401-508   per_mask_trafo_from_pointwise_flows /
 *                                        weighted_pc_alignment_for_different_batched_slotted_kabsch_weights
 *   liso/torch_symm_ortho/__init__.py:7-87  SymmetricOrthogonalization forward (R = U Vh of an fp64 SVD) + backward
 * The reference materialises [B,S,N,4] point-in-box coordinates and four [B,S,N,3] tensors; here one pass over the
 * points produces all per-slot weighted moments, and a 3x3 one-sided Jacobi (no library SVD) gives U, D, Vh.
 *
 * All pointers are device pointers; nothing allocates or synchronises; returns LISO_OK or a negative code.
 */
#ifndef LISO_KABSCH_H
#define LISO_KABSCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISO_KABSCH_NMOM 9 /* sum w | sum w x (2) | sum w y (2) | sum w y x^T (4), x relative to the slot's box centre */

typedef struct {
    int batch, n_points, n_slots; /* B, N (padded length), S foreground slots */
    int point_stride;             /* floats between consecutive points of `points` (>= 3) */
    int flow_stride;              /* floats between consecutive points of `flow` (>= 2, only x,y are read) */
    float slope;                  /* mask_rendering.pred_sigmoid_slope (15) */
    float scale_fg, scale_bg;     /* 1 - obj_dim_scale_buffer, 1 + obj_dim_scale_buffer (0.75, 1.25) */
    int softness;                 /* 0 = cauchy (0.5 + atan(x)/pi), 1 = sigmoid */
} liso_kabsch_cfg;

/* scratch for liso_kabsch_trafos_f32 */
size_t liso_kabsch_workspace_bytes(const liso_kabsch_cfg* cfg);

/* points [B,N,point_stride] (padding rows may hold NaN), valid [B,N] (uint8), flow [B,N,flow_stride],
 * box_pos [B,S,3], box_dims [B,S,3], box_rot [B,S] ->
 *   trafos   float64 [B, S+1, 4, 4]  (foreground slots then the background slot)
 *   cum_wts  float32 [B, S+1]
 *   fg_weights float32 [B, S, N] or NULL (kabsch_mask.py:353-360; invalid points get weight 0, :417-419)
 * Semantics: z of the aligned clouds is zeroed (:413), background weight = 1 - screen(fg masks at scale_bg)
 * (:362-372), slots whose total weight is < 1e-12 fall back to uniform weights 1e-12 (:452-470),
 * R = U Vh without reflection fix (:489-492). */
int liso_kabsch_trafos_f32(const liso_kabsch_cfg* cfg, const float* points, const uint8_t* valid, const float* flow,
                           const float* box_pos, const float* box_dims, const float* box_rot, double* trafos,
                           float* cum_wts, float* fg_weights, void* workspace, size_t workspace_bytes, void* stream);

/* The same for callers with a FIXED number of slots of which only the first slot_count[b] (int32 [B], device) hold boxes (the LISO
 * loop's box mining: 64 slots, 15-30 boxes; the other slots are parked where their soft mask is 0 in fp32).  The parked slots are not
 * evaluated -- 7.7 M box-mask evaluations (3 atan each) per 120k-point cloud at 64 slots, the kernel is bound by them -- and the lanes
 * are re-dealt over the boxes that exist (16 / 32 / 64 slot lanes x 4 / 2 / 1 point sub-lanes per wave, chosen from slot_count[b]).
 * Skipping a parked slot changes no output bit (it is a factor of exactly 1 in the background product and its moments fall under the
 * 1e-12 rule above; fg_weights of parked slots: exact zeros instead of ~1e-23), but the re-deal changes the ORDER in which a slot's
 * fp32 moment sums are added: the transforms equal liso_kabsch_trafos_f32's for the parked arrangement to fp32-summation tolerance
 * (1e-6 relative, what tests/test_gpu_kabsch.py asserts), not bitwise, and they depend bitwise on which of the three deals
 * slot_count[b] selects.  For a fixed slot_count every output is run-to-run bitwise reproducible.
 * slot_count == NULL: all n_slots slots hold boxes; the deal is chosen from n_slots, so liso_kabsch_trafos_f32 on exactly k boxes
 * and this call on more slots with slot_count[b] = k use the same deal and give the same bits for those boxes. */
int liso_kabsch_trafos_counted_f32(const liso_kabsch_cfg* cfg, const float* points, const uint8_t* valid, const float* flow,
                                   const float* box_pos, const float* box_dims, const float* box_rot, const int32_t* slot_count,
                                   double* trafos, float* cum_wts, float* fg_weights, void* workspace, size_t workspace_bytes,
                                   void* stream);

/* SymmetricOrthogonalization.forward for n 3x3 matrices (row-major fp64): R = U Vh; U, Vh, D are saved for backward. */
int liso_symm_ortho_fwd_f64(const double* a, int n, double* r, double* u, double* vh, double* d, void* stream);

/* SymmetricOrthogonalization.backward (torch_symm_ortho/__init__.py:15-43): grad_A = U (W - W^T) Vh with
 * W_kl = (U^T grad_R V)_kl / (d_k + d_l + delta_kl). */
int liso_symm_ortho_bwd_f64(const double* grad_r, const double* u, const double* vh, const double* d, int n,
                            double* grad_a, void* stream);

/* ---- weighted moments of a differentiable weighted Kabsch fit ------------------------------------------------------
 * Replaces the torch reductions of weighted_pc_alignment (liso/slim/slim_loss/weighted_pc_alignment.py:36-47; same
 * code in liso/weighted_pc_alignment/weighted_pc_alignment.py:80-110): two weighted means, two centred clouds and the
 * 3xN . Nx3 product -- 12 launches forward, ~25 backward and a skinny GEMM per call, 12 calls per SLIM step.  Here
 * the clouds are read once and 16 fp64 sums come back:
 *     out[0]      = sum_i w_i                out[1..3] = sum_i w_i x_i         out[4..6] = sum_i w_i y_i
 *     out[7+3a+b] = sum_i w_i y_i[a] x_i[b]
 * from which the host forms  m_x = S_x/S, m_y = S_y/S, S_xy = (S_yx - S m_y m_x^T)/S  (== the reference's centred
 * product) with differentiable 3x3 fp64 algebra.  Batched over `batch` independent fits of the same length:
 * x, y: float32 [batch,n,3] (finite); w: float32 [batch,n] (rows to ignore: 0); out float64 [batch,16].
 * Backward: grad_out float64 [batch,16] -> grad_x/grad_y [batch,n,3], grad_w [batch,n] (any of them may be NULL). */
size_t liso_weighted_moments_workspace_bytes(int batch);
int liso_weighted_moments_fwd_f32(const float* x, const float* y, const float* w, int batch, long n, double* out, void* workspace,
                                  size_t workspace_bytes, void* stream);
int liso_weighted_moments_bwd_f32(const float* x, const float* y, const float* w, int batch, long n, const double* grad_out,
                                  float* grad_x, float* grad_y, float* grad_w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_KABSCH_H */
