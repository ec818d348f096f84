/*
 * liso_optim.h -- C ABI of the detector's parameter update (gfx950): decoupled-weight-decay Adam over ONE flat fp32 buffer.
 *
 * Replaces torch.optim.AdamW.step() as the reference calls it in the detector train step
 *   liso/liso_cli.py:615-618   loss.backward(); optimizer.step(); lr_scheduler.step()
 *   liso/liso_cli.py:792-823   AdamW(box_predictor.parameters(), lr, weight_decay=0.01) driven by OneCycleLR (which rewrites
 *                              lr AND beta1 of the parameter group before every step)
 * which PyTorch runs as ~12 multi-tensor launches over ~100 parameter tensors (1.2 ms of host time per step on this path,
 * measured).  Here parameters, gradients and both moments live in four flat buffers with identical element order (the
 * trainer makes every nn.Parameter / .grad a strided view into them): the update is one HBM-bound pass,
 * 16 B read + 12 B written per element.
 *
 * Arithmetic per element (fp32, the operation order of torch/optim/adamw.py `_multi_tensor_adamw`, maximize = amsgrad = 0):
 *   p   <- p * (1 - lr * weight_decay)
 *   m   <- m + (1 - beta1) * (g - m)
 *   v   <- v * beta2 + (1 - beta2) * g * g
 *   p   <- p - (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * `step` is the 1-based count of THIS update.  The scalar factors are evaluated in double on the host side of the call.
 * All pointers are device pointers; nothing allocates or synchronises; the call enqueues on `stream` and returns LISO_OK or
 * a negative LISO_E* code (include/liso_iou3d.h).  Graph-capturable (the scalars are then baked into the captured node).
 */
#ifndef LISO_OPTIM_H
#define LISO_OPTIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int liso_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                        double beta2, double eps, double weight_decay, long step, void* stream);

/* The same update on grad * grad_scale: data-parallel training all-reduces the flat gradient buffer with SUM and passes
 * grad_scale = 1 / world_size here instead of launching a division over the buffer (the reference is single-GPU; the gradient mean
 * over ranks is this build's addition).  grad_scale = 1 is bit-identical to liso_adamw_step_f32. */
int liso_adamw_step_scaled_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                               double beta2, double eps, double weight_decay, double grad_scale, long step, void* stream);

/* RMSprop over one flat fp32 buffer (SLIM's optimizer, liso/slim/experiment.py:200-219: torch.optim.RMSprop(lr) with its defaults
 * alpha 0.99, eps 1e-8, no momentum, not centered), the element-wise operations of torch's multi-tensor implementation:
 *     square_avg = square_avg * alpha + (1 - alpha) * g * g;   param = param - lr * g / (sqrt(square_avg) + eps),   g = grad * grad_scale
 * All buffers 16-byte aligned; one launch. */
int liso_rmsprop_step_f32(float* param, const float* grad, float* square_avg, size_t n, double lr, double alpha, double eps,
                          double grad_scale, void* stream);

/* `count` fp32 arrays copied into their destinations by ONE launch per LISO_GATHER_MAX jobs: dst[k][0 .. numel[k]) = src[k][...].
 * The detector step uses it for the parameter gradients that autograd hands back as tensors of their own (merged / sliced
 * parameters whose gradient no kernel can write in place): with `.grad = None` autograd keeps those tensors instead of launching one
 * `add_` per parameter into the zeroed flat buffer (torch/csrc/autograd/functions/accumulate_grad.h), and this call moves all of
 * them into their flat-buffer slices.  src / dst / numel are HOST arrays of device pointers / element counts (baked into the launch). */
#define LISO_GATHER_MAX 48
int liso_gather_f32(int count, const void* const* src, void* const* dst, const size_t* numel, void* stream);

/* `n` device-to-device copies of any type in ONE launch per LISO_MULTI_COPY_MAX segments: dst[k][0 .. bytes[k]) = src[k][...]
 * (16-byte vectors where both pointers allow).  dst / src / bytes are HOST arrays (baked into the launch); segments must not overlap.
 * The training steps stage the inputs of their captured hipGraphs with it (10 target tensors per detector step, the cloud tensors of a
 * SLIM inference replay): one launch instead of one runtime buffer copy per tensor. */
#define LISO_MULTI_COPY_MAX 24
int liso_multi_copy(int n, void* const* dst, const void* const* src, const size_t* bytes, void* stream);

/* `n` 2-D copies in one launch per LISO_MULTI_COPY_ROWS_MAX jobs: job k copies rows[k] rows of row_bytes[k] contiguous bytes,
 * dst_stride[k] / src_stride[k] bytes between consecutive rows (everything a multiple of 4).  The merged filters of parallel
 * convolutions (block-diagonal / concatenated / channel-permuted: liso/slim/model/update.py:29-164 and
 * liso/networks/centerpoint/center_head.py:60-117 run their parallel branches as single launches) are assembled from the modules' own
 * parameters -- and their gradients handed back -- by one such launch instead of one framework copy per block. */
#define LISO_MULTI_COPY_ROWS_MAX 32
int liso_multi_copy_rows(int n, void* const* dst, const void* const* src, const unsigned* rows, const unsigned* row_bytes,
                         const size_t* dst_stride, const size_t* src_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LISO_OPTIM_H */
