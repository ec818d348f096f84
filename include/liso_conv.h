/*
 * liso_conv.h -- C ABI of the MI355X-native 2-D convolutions of the BEV networks (gfx950 MFMA implicit GEMM).
 *
 * Replaces the cuDNN convolutions (+ the BatchNorm-apply / bias / ReLU kernels around them) the reference reaches
 * through torch.nn.Conv2d / ConvTranspose2d in
 *   liso/networks/centerpoint/rpn.py:113-146            (SECOND-style backbone: 3x3 s1/s2, 1x1, 2x2 s2, transposed 2x2 s2)
 *   liso/networks/centerpoint/center_head.py:60-117     (shared 3x3 conv, SepHead 3x3 convs)
 *   liso/slim/model/update.py:96-164                    (RAFT update block: 1x1, 3x3, 7x7, ConvGRU)
 *   liso/slim/model/extractor.py:211-297                (SmallEncoder: 7x7 s2, residual 3x3 s1/s2, 1x1)
 * forward, data gradient and weight gradient.
 *
 * One descriptor covers all of them.  Tensors are NHWC ("channels last": [B, H, W, C], C contiguous; a pixel stride
 * larger than C addresses a channel slice of a wider tensor).  The kernels work on a *virtual grid* (hv x wv): virtual
 * pixel v reads input pixels v*is + (tap_dy, tap_dx) and writes output pixel v*os + (class_ooy, class_oox).
 *   forward conv, stride s, padding p:    is = s, os = 1, taps (kh - p, kw - p), one class
 *   data gradient of a stride-1 conv:     the same with mirrored taps and transposed weights
 *   data gradient of a stride-2 conv /    os = 2: up to 4 output-parity classes, each with the subset of taps that
 *   forward of a transposed conv:         reaches it (no multiplications by inserted zeros)
 *
 * Arithmetic: `mode` LISO_CONV_BF16 -- bf16 tensors, fp32 accumulation on v_mfma_f32_32x32x16_bf16;
 *             `mode` LISO_CONV_F32X3 -- fp32 tensors; every operand is split on the fly into bf16 hi + lo parts and each
 *             product is evaluated as hi*hi + hi*lo + lo*hi (three MFMAs, fp32 accumulation): relative error per product
 *             <= 2^-16 (fp32: 2^-24; TF32, cuDNN's default for fp32 convolutions on Ampere: 2^-11) at 3/16 of the cost
 *             of the native fp32 MFMA.
 *             `mode` LISO_CONV_F32 -- fp32 tensors, EXACT fp32 arithmetic on v_mfma_f32_32x32x2_f32 (fp32 operands, fp32
 *             accumulation: bit for bit a k-ordered fmaf chain, 2^-24 per product like the reference's fp32 path; 157 TFLOP/s
 *             peak): the parity configuration the gradient tests against the reference fixtures run in.
 * Fusions (all optional): prologue x' = relu?(x * in_scale[c] + in_shift[c]) applied while the input tile is staged
 * (the BatchNorm-apply + ReLU of the producing layer: normalised activations never touch HBM; padding stays exactly 0);
 * epilogue: + bias[c], ReLU, per-channel partial sums of the stored values for the BatchNorm statistics of THIS layer.
 *
 * Packed weights (liso_conv_pack_weights): [plane][tap][ci_pad / 8][co_pad][8] bf16, plane 0 = hi (or the bf16 value),
 * plane 1 = lo (F32X3 only); ci_pad = round_up(ci, 16), co_pad = round_up(co, 64); padding is zero.
 * LISO_CONV_F32: [tap][ci_pad / 4][co_pad][4] fp32 (unrounded), the same number of bytes as the two F32X3 planes.
 *
 * All pointers are device pointers; nothing allocates or synchronises; every call enqueues on `stream` and returns
 * LISO_OK or a negative LISO_E* code (include/liso_iou3d.h).  Graph-capturable.
 */
#ifndef LISO_CONV_H
#define LISO_CONV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISO_CONV_MAX_TAPS 49
#define LISO_CONV_MAX_CLASSES 4
#define LISO_CONV_BF16 0
#define LISO_CONV_F32X3 1
#define LISO_CONV_F32 2

typedef struct {
    int batch, hi, wi, ci;      /* input  [batch, hi, wi, ci] */
    int x_pix_stride;           /* elements between consecutive input pixels (>= ci) */
    int ho, wo, co;             /* output [batch, ho, wo, co] */
    int y_pix_stride, y_ch_off; /* elements between output pixels; first output channel inside the pixel row */
    int hv, wv;                 /* virtual grid (per class) */
    int isy, isx, osy, osx;
    int n_classes;
    int class_tap_begin[LISO_CONV_MAX_CLASSES + 1]; /* taps of class c: [begin[c], begin[c + 1]) */
    int class_ooy[LISO_CONV_MAX_CLASSES], class_oox[LISO_CONV_MAX_CLASSES];
    int n_taps;                 /* over all classes */
    int tap_dy[LISO_CONV_MAX_TAPS], tap_dx[LISO_CONV_MAX_TAPS]; /* input offset of the tap */
    int tap_w[LISO_CONV_MAX_TAPS];                              /* tap index inside the packed weights */
    int w_taps;                 /* taps in the packed weights (kh * kw) */
    int mode;                   /* LISO_CONV_BF16 | LISO_CONV_F32X3 | LISO_CONV_F32 */
    int out_f32;                /* BF16 mode: 1 = fp32 output, 0 = bf16 output (F32X3: always fp32) */
    int in_relu, out_relu;
    int in_affine_batch_stride; /* 0: in_scale / in_shift are [ci], shared by all samples (BatchNorm);
                                   > 0: per-sample vectors, sample b reads in_scale[b * stride + c] (InstanceNorm) */
    int wgrad_co;               /* liso_conv_wgrad only: output channels actually written to dw / dbias (0 = co).  dy may carry
                                   zero channels beyond it (a 1-3 channel gradient padded to one 16-B group): the extra filters
                                   are computed and dropped, dw / dbias keep the layer's true shape */
} liso_conv_desc;

/* Packs torch-layout fp32 weights for the kernels.
 *   transposed == 0: src[d0][d1][kh][kw] = weight of nn.Conv2d  (d0 = out channels, d1 = in channels)
 *   transposed == 1: the same memory read as nn.ConvTranspose2d (d0 = in channels, d1 = out channels)
 *   for_dgrad  != 0: pack for the data-gradient launch (the roles of in / out channels are exchanged)
 * dst: bf16 [planes][kh*kw][K_pad/8][N_pad][8], planes = 1 (BF16) or 2 (F32X3), or fp32 [kh*kw][K_pad/4][N_pad][4] (F32);
 * bytes = liso_conv_packed_bytes(). */
size_t liso_conv_packed_bytes(int k_channels, int n_channels, int taps, int mode);
int liso_conv_pack_weights(const float* src, int d0, int d1, int kh, int kw, int transposed, int for_dgrad, int mode,
                           void* dst, void* stream);

/* The same for several weight tensors in ONE launch (a training step packs every layer twice -- forward and data-gradient
 * panels: 57 launches of ~5 us for the CenterPoint backbone + head).  `jobs` is a HOST array read during the call. */
typedef struct {
    const float* src;
    void* dst;
    int d0, d1, kh, kw, transposed, for_dgrad, mode;
} liso_conv_pack_job;
int liso_conv_pack_weights_batched(const liso_conv_pack_job* jobs, int n_jobs, void* stream);

/* rows of the statistics buffer one forward launch writes: stats_partial is fp32 [rows][2][co_pad]
 * (sum and sum of squares of (stored value - stats_shift[c]) over the pixels of one block). */
int liso_conv_stats_rows(const liso_conv_desc* d);

/* which kernel liso_conv_forward launches for `d`: 0 = conv_igemm_kernel, 1 = conv_roles_kernel (3x3 / stride 1 / one tap class: loader
 * waves + MFMA waves, persistent blocks), 2 = conv_1x1_kernel (one tap, fp32 tensors: fragments straight from global memory),
 * 3 = conv_taps_kernel (any window on <= 8 fp32 input channels: two taps per MFMA step), -1 = geometry not covered.  For measurement
 * code that attributes launch times to kernels (bench.py's `roofline`) and for tests; the choice itself is internal. */
int liso_conv_kernel_kind(const liso_conv_desc* d);

/* the launch plan of liso_conv_forward for `d` (tests, measurement scripts): info = {kernel kind as above, tile rows / 4, panel width / 32,
 * wave groups that share the channel slabs of a block (conv_igemm_kernel; 1 otherwise), blocks (conv_roles_kernel: tiles, walked by at
 * most one persistent block per compute unit), LDS bytes per block, channels per slab, taps per weight stage} */
int liso_conv_plan_info(const liso_conv_desc* d, int info[8]);

/* y = conv(x') (+ bias) ; bias / in_scale / in_shift / stats_partial / stats_shift may be NULL */
int liso_conv_forward(const liso_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                      const float* in_scale, const float* in_shift, void* y, float* stats_partial,
                      const float* stats_shift, void* stream);

/* The same for a SPARSE input: `occupancy` fp32 [batch, hi, wi] with 0 where the input pixel is exactly zero in every channel
 * (the pillar canvas of liso/networks/pcl_to_feature_grid, whose occupancy map PointPillarsScatter produces beside it,
 * pillar_scatter.py:62-102): blocks whose whole input window is unoccupied skip their loads and multiplications -- they
 * would only add exact zeros -- and write bias / ReLU / statistics like every other block.  Results are bit-identical to
 * liso_conv_forward.  Requires in_scale == NULL (a prologue would turn the zeros into relu(shift)).  occupancy == NULL: dense. */
int liso_conv_forward_sparse(const liso_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                             const float* in_scale, const float* in_shift, void* y, float* stats_partial,
                             const float* stats_shift, const float* occupancy, void* stream);

/* Weight gradient for a SPARSE fp32 input (the SLIM encoders' 7x7 / 2 stem on the pillar canvas, liso/slim/model/extractor.py:230-232
 * with the canvas of pillar_scatter.py:62-102): `occupancy` fp32 [batch, hi, wi], 0 where the input pixel is exactly zero in every
 * channel.  Only occupied cells are visited (listed on the device in (sample, row, column) order), products and sums are exact fp32
 * FMAs in a fixed order (bitwise reproducible; at least the accuracy of either fp32 mode of liso_conv_wgrad).  `d` = the forward
 * descriptor of a one-class (gather-form) convolution with ci, co <= 64 and no prologue; dw / dbias as liso_conv_wgrad
 * (transposed = 0).  0 bytes from the workspace query = geometry not covered. */
size_t liso_conv_wgrad_sparse_workspace_bytes(const liso_conv_desc* d);
int liso_conv_wgrad_sparse_f32(const liso_conv_desc* d, const float* x, const float* occupancy, const float* dy, int dy_pix_stride,
                               float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient of a k x k (k = 3 | 5 | 7), stride-1, padding k/2 convolution whose fp32 input has at most FOUR channels: the motion
 * encoder's `conv_flow1` / `conv_class1` on the flow / the class logits (liso/slim/model/update.py:57,66; autograd's
 * convolution_backward for the weights in the reference).  x: NHWC with 4 stored channels per pixel (a 2- or 3-channel input is
 * zero-padded by the caller; the gradient rows of the padding are zeros), pixel stride `x_pix_stride` floats (a multiple of 4, 16-B
 * aligned base); dy: NHWC [batch, h, w, co], pixel stride `dy_pix_stride`.  dw fp32 [co][4][k][k] (torch's layout), dbias [co] or
 * NULL.  Exact fp32 FMAs, fixed summation order.  0 bytes from the workspace query = geometry not covered. */
size_t liso_conv_wgrad_smallci_workspace_bytes(int batch, int h, int w, int co, int k);
int liso_conv_wgrad_smallci_f32(const float* x, long x_pix_stride, const float* dy, long dy_pix_stride, int batch, int h, int w, int co,
                                int k, float* dw, float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient of the convolution described by `d` (a FORWARD descriptor: x = layer input, with the same optional
 * prologue, dy = gradient of the layer output [batch, ho, wo, co] with pixel stride dy_pix_stride):
 *   dw[co][ci][kh][kw] (torch layout of nn.Conv2d; transposed != 0: [ci][co][kh][kw]) = sum over pixels, overwritten;
 *   dbias[co] = sum over pixels of dy (may be NULL).
 * workspace: liso_conv_wgrad_workspace_bytes(d) bytes of device scratch (split-K slabs). */
size_t liso_conv_wgrad_workspace_bytes(const liso_conv_desc* d);
int liso_conv_wgrad(const liso_conv_desc* d, const void* x, const float* in_scale, const float* in_shift, const void* dy,
                    int dy_pix_stride, int transposed, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                    void* stream);

/* BatchNorm statistics from the partial sums of a forward launch (fixed order, fp64 merge):
 *   mean = shift + S1 / n, var = S2 / n - (S1 / n)^2 (biased), n = batch * ho * wo
 *   stats[4 * c] = scale | shift | mean | invstd with scale = gamma * invstd, shift = beta - mean * scale
 *   (the layout liso_bn_relu_bwd of include/liso_bn.h consumes);
 *   running_mean / running_var (unbiased) are updated in place with `momentum` when they are non-NULL. */
int liso_conv_bn_finalize(const float* stats_partial, int rows, int co, int co_pad, long n, const float* stats_shift,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, float* stats, void* stream);

/* InstanceNorm2d statistics from the partial sums of a forward launch of a one-class (gather) descriptor, whose rows are ordered
 * sample-major: per sample b and channel c over the ho * wo pixels of that sample
 *   stats[b][4 * co] = scale | shift | mean | invstd, scale = gamma[c] * invstd, shift = beta[c] - mean * scale
 *   (gamma / beta may be NULL: 1 / 0, nn.InstanceNorm2d's default affine=False); biased variance, like torch.
 * rows_per_sample = liso_conv_stats_rows(d) / d->batch.
 * Replaces the InstanceNorm2d layers of liso/slim/model/extractor.py:5-71,211-297 (norm_fn "instance"). */
int liso_conv_in_finalize(const float* stats_partial, int rows_per_sample, int batch, int co, int co_pad, long n_per_sample,
                          const float* gamma, const float* beta, float eps, float* stats, void* stream);

/* out = relu( fa(a) + fb(b) ) on NHWC fp32 tensors [batch, pixels, c], c % 4 == 0, where
 *   fa(a) = a                                   if a_scale == NULL
 *         = relu?(a * a_scale[n * a_stride + c] + a_shift[n * a_stride + c])  otherwise (per-sample vectors, a_stride elements
 *           apart, 16-byte aligned; a_relu selects the ReLU)
 * and fb likewise: the tail of a residual block -- relu(x + y) with the pending normalisation (+ ReLU) of either branch applied
 * on the fly (liso/slim/model/extractor.py:29-38: `self.relu(x + y)`, y = relu(norm2(conv2(.))), x = norm3(conv1x1(.)) or the
 * block input).  One read of each input, one write. */
int liso_residual_affine_relu_f32(const float* a, const float* a_scale, const float* a_shift, int a_stride, int a_relu, const float* b,
                                  const float* b_scale, const float* b_shift, int b_stride, int b_relu, float* out, int batch,
                                  long pixels, int c, void* stream);

/* Process-wide launch-plan options.  LISO_CONV_OPT_SHARED_GPU (value != 0): kernels of other streams run next to the convolutions
 * (the three pipeline stages of the LISO loop) -- plans then never spend a CU's whole LDS on one block, so that other kernels' blocks
 * can share the CU.  Results are unaffected.  Returns LISO_EINVAL for an unknown option. */
#define LISO_CONV_OPT_SHARED_GPU 1
/* LISO_CONV_OPT_ROLES_CUS (value = n >= 8; 0 = all): the persistent blocks of conv_roles_kernel launches planned from now on occupy at
 * most n compute units (one block each).  With LISO_INFER_CUS=128 the LISO loop plans the frozen SLIM's inference launches (captured into
 * hipGraphs: the grid is baked in) that way: the detector step on the other stream -- the pipeline's critical path, a chain of ~250
 * small dependent kernels -- then always finds free CUs instead of waiting for a 60-us launch that holds all 256 to drain (measured:
 * 4.16-4.17 vs 4.23-4.26 ms per step).  Results are unaffected. */
#define LISO_CONV_OPT_ROLES_CUS 2
int liso_conv_set_option(int option, int value);

/* ---- sparse form of the SLIM encoders' first convolution --------------------------------------------------------------------------
 * y = relu?(conv7x7 / stride 2 / padding 3, 64 -> 32 channels (x) + bias) for an fp32 pillar canvas `x` (NHWC [batch, hi, wi, 64],
 * pixel stride `x_pix_stride` floats) with its occupancy map (fp32 [batch, hi, wi], 0 = the cell is exactly zero in every channel):
 * liso/slim/model/extractor.py:230-232,283-286 on the canvas of pillar_scatter.py:62-102.  Only the occupied cells are multiplied
 * (F32X3 arithmetic on the matrix cores, `w_packed` = liso_conv_pack_weights(..., LISO_CONV_F32X3) of the [32, 64, 7, 7] filter);
 * every output pixel [batch, hi / 2, wi / 2, 32] is written once.  `stats_partial` != NULL: per-block sums / sums of squares of the
 * output, [blocks][2][32] floats with blocks = batch * (hi / 2) * (wi / 2) / (128 * liso_sparse_conv_stat_groups(hi, wi, 32)), for
 * liso_conv_in_finalize (rows_per_sample = blocks / batch, co_pad = 32).  `max_cells_per_sample`: capacity of the cell lists (the voxeliser's max_voxels); if a batch holds more occupied
 * cells than batch * max_cells_per_sample the surplus is dropped and *overflow (device int, may be NULL) is set to 1.
 * hi even, wi a multiple of 64.  Same results as liso_conv_forward up to the fp32 summation order. */
/* The general entry points behind it: k x k (k = 3 | 7), stride 2, padding k / 2, 64 input channels, co = 64 (k = 3: the detector's
 * first RPN layer, liso/networks/centerpoint/rpn.py:113-131 on the canvas of pillar_scatter.py) or 32 (k = 7: the SLIM stem); bf16
 * tensors (is_bf16, one MFMA per product) or fp32 tensors in F32X3 arithmetic; y has the dtype of x; `stats_partial` [blocks][2][co]
 * holds sums of (y - stats_shift[c]) and their squares (y as stored) for liso_conv_bn_finalize / liso_conv_in_finalize, blocks =
 * batch * (hi / 2) * (wi / 2) / ((4096 / co) * liso_sparse_conv_stat_groups(hi, wi, co)).  liso_sparse_conv_dgrad: the data gradient of that convolution AT THE OCCUPIED CELLS
 * (dx rows of other cells are not written: the caller zero-fills dx; the pillar encoder's backward reads occupied cells only);
 * `w_packed_dgrad` = liso_conv_pack_weights(..., for_dgrad = 1, ...).  Workspace query: for_dgrad != 0 omits the product buffer.
 * reuse_lists != 0: `workspace` is the (unmodified) workspace of the liso_sparse_conv_forward call on the same canvas -- its cell
 * lists are used as they are (occupancy may be NULL) instead of being rebuilt. */
int liso_sparse_conv_stat_groups(int hi, int wi, int co);
size_t liso_sparse_conv_workspace_bytes(int batch, int hi, int wi, int k, int co, int max_cells_per_sample, int for_dgrad);
int liso_sparse_conv_forward(const void* x, long x_pix_stride, int is_bf16, const float* occupancy, const void* w_packed,
                             const float* bias, int batch, int hi, int wi, int k, int co, int max_cells_per_sample, int relu, void* y,
                             float* stats_partial, const float* stats_shift, int* overflow, void* workspace, size_t workspace_bytes,
                             void* stream);
int liso_sparse_conv_dgrad(const void* dy, long dy_pix_stride, int is_bf16, const float* occupancy, const void* w_packed_dgrad, int batch,
                           int hi, int wi, int k, int co, int max_cells_per_sample, void* dx, long dx_pix_stride, int* overflow,
                           void* workspace, size_t workspace_bytes, int reuse_lists, void* stream);
size_t liso_sparse_stem_workspace_bytes(int batch, int hi, int wi, int max_cells_per_sample);
int liso_sparse_stem_forward_f32(const float* x, long x_pix_stride, const float* occupancy, const void* w_packed, const float* bias,
                                 int batch, int hi, int wi, int max_cells_per_sample, int relu, float* y, float* stats_partial,
                                 int* overflow, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_CONV_H */
