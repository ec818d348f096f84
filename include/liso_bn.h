/*
 * liso_bn.h -- C ABI of the fused BatchNorm2d(+ReLU) for channels-last BEV feature maps (gfx950).
 *
 * Replaces the norm + activation pairs of the reference's BEV backbone / head
 *   liso/networks/centerpoint/rpn.py:113-131 (`BatchNorm2d` + `ReLU` after every conv; built by norm.py:55-56)
 *   liso/networks/centerpoint/center_head.py:36-38,81-92
 * which PyTorch runs as 3 (forward) + 3 (backward) BatchNorm launches plus separate ReLU kernels.  Here:
 *   forward  = stats (1 read of x) -> finalize (tiny) -> apply+ReLU (1 read, 1 write)
 *   backward = reduce (read dy, x) -> finalize (tiny) -> dx (read dy, x; write dx), ReLU mask recomputed from x
 * Batch statistics use block-shifted sums merged over the block means in fp64 (fixed order):
 * no E[x^2]-E[x]^2 cancellation (MIOpen's spatial BN loses ~2e-4 relative at mean/std = 50, measured), reproducible.
 *
 * x, y, dy, dx: [M, C] row-major (M = N*H*W pixels, channels-last), dtype fp32 (is_bf16 = 0) or bf16 (is_bf16 = 1);
 * C % 8 == 0, C <= 256 (V = 4 channels per lane in fp32, 8 in bf16; 256 / (C / V) rows per block pass).  gamma/beta/running_*: fp32 [C].  stats: fp32 [4*C] = scale | shift | mean | invstd.
 * All pointers are device pointers; nothing allocates or synchronises.
 */
#ifndef LISO_BN_H
#define LISO_BN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bytes of device scratch for the calls below (depends only on C) */
size_t liso_bn_workspace_bytes(int c);

/* training != 0: batch statistics (biased variance for normalisation, unbiased for running_var, torch semantics),
 * running stats updated in place with `momentum`;  training == 0: running statistics.  relu != 0 fuses max(.,0). */
int liso_bn_relu_fwd(const void* x, int is_bf16, long m, int c, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, int training, int relu,
                     void* y, float* stats, void* workspace, size_t workspace_bytes, void* stream);

/* grad_gamma/grad_beta [C] fp32 overwritten; dx has the dtype of x. */
int liso_bn_relu_bwd(const void* dy, const void* x, int is_bf16, long m, int c, const float* gamma, const float* stats,
                     int training, int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace,
                     size_t workspace_bytes, void* stream);

/* The same backward on rows that are CHANNEL SLICES of wider channels-last tensors: row r of dy / x / dx starts dy_stride / x_stride /
 * dx_stride elements after row r - 1 (>= C, multiples of 16 bytes; the pointers address the slice's first channel).  The BatchNorms
 * behind a channel concatenation (the three deblocks in front of the head, liso/networks/centerpoint/rpn.py:140-146) then read and
 * write the concatenated tensors in place: no slice copies, no concatenation of the partial input gradients. */
int liso_bn_relu_bwd_strided(const void* dy, long dy_stride, const void* x, long x_stride, int is_bf16, long m, int c, const float* gamma,
                             const float* stats, int training, int relu, void* dx, long dx_stride, float* grad_gamma, float* grad_beta,
                             void* workspace, size_t workspace_bytes, void* stream);

/* The same backward in TWO launches instead of three: the last block of the reduction to finish also turns the partial sums into
 * grad_gamma / grad_beta / the dx coefficients.  `ticket`: one device-resident unsigned that is ZERO on entry and zero again on
 * return (the caller keeps one per BatchNorm layer, zeroed once; two calls in flight at the same time must not share it). */
int liso_bn_relu_bwd_ticket(const void* dy, const void* x, int is_bf16, long m, int c, const float* gamma, const float* stats,
                            int training, int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace,
                            size_t workspace_bytes, unsigned* ticket, void* stream);

/* ---- InstanceNorm2d(+ReLU), training: the same passes with one set of statistics per sample ---------------------------------
 * Replaces `nn.InstanceNorm2d(affine=True)` + `ReLU` of the SLIM encoders in training (liso/slim/model/extractor.py:24-38,
 * 219-230; norm_fn "instance" / "instance_affine"), which PyTorch runs through the BatchNorm kernels on a [1, B*C, H, W] view
 * (weight / bias repeated B times) in NCHW.  x, y, dy, dx: [groups, m, C] channels-last (groups = samples, m = H*W pixels each);
 * stats fp32 [groups, 4*C]; grad_gamma / grad_beta fp32 [groups, C] (per sample: the caller adds the samples).
 * No running statistics (track_running_stats = False). */
size_t liso_in_workspace_bytes(int groups, int c);
int liso_in_relu_fwd(const void* x, int is_bf16, int groups, long m, int c, const float* gamma, const float* beta, float eps, int relu,
                     void* y, float* stats, void* workspace, size_t workspace_bytes, void* stream);
int liso_in_relu_bwd(const void* dy, const void* x, int is_bf16, int groups, long m, int c, const float* gamma, const float* stats,
                     int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes, void* stream);
/* The same backward with grad_gamma / grad_beta fp32 [C] = the SUM over the samples (what the shared affine parameters receive),
 * added in sample order inside the finalize launch: no [groups, C] intermediate, no reduction launches behind the call. */
int liso_in_relu_bwd_sum(const void* dy, const void* x, int is_bf16, int groups, long m, int c, const float* gamma, const float* stats,
                         int relu, void* dx, float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_BN_H */
