/*
 * liso_iou3d.h -- C ABI of the MI355X-native rotated-BEV IoU / NMS ops.
 *
 * Drop-in boundary for the reference's pybind module `iou3d_nms_cuda`
 * (reference: iou3d_nms/src/iou3d_nms_api.cpp:11-17, prototypes
 * iou3d_nms/src/iou3d_nms.h:9-12 and iou3d_nms/src/iou3d_cpu.h:9).
 * The reference binds at::Tensor; this ABI takes plain device pointers and a
 * hipStream_t (as void*) so that any host language can bind it.  The Python
 * shim that restores the five pybind names lives in liso_amd/iou3d_nms_cuda.py.
 *
 * Conventions
 *   - boxes are float32 rows of 7: [x, y, z, dx, dy, dz, heading], contiguous.
 *   - all *device* pointers must be valid on the current HIP device.
 *   - no allocation, no host synchronisation, no exit(): every entry point
 *     enqueues work on `stream` and returns 0, or a negative LISO_E* code.
 *     (The reference prints and calls exit(-1): iou3d_nms.cpp:14-38.  Returning
 *     an error code is an intentional deviation.)
 *   - graph-capturable: nothing here calls hipMalloc/hipFree/hipMemcpy.
 */
#ifndef LISO_IOU3D_H
#define LISO_IOU3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LISO_OK 0
#define LISO_EINVAL (-1)    /* null pointer / negative size */
#define LISO_EWORKSPACE (-2) /* workspace too small */
#define LISO_ELAUNCH (-3)   /* hipGetLastError() != hipSuccess after launch */

/* replaces boxes_overlap_bev_gpu (iou3d_nms.cpp:49-68; kernel iou3d_nms_kernel.cu:236-249).
 * out[n*m] <- area of the intersection polygon of a[i] and b[j]. */
int liso_iou3d_overlap_bev_f32(const float* boxes_a, int n, const float* boxes_b, int m, float* out, void* stream);

/* replaces boxes_iou_bev_gpu (iou3d_nms.cpp:70-88; kernel iou3d_nms_kernel.cu:251-265). */
int liso_iou3d_iou_bev_f32(const float* boxes_a, int n, const float* boxes_b, int m, float* out, void* stream);

/* bytes of device scratch liso_iou3d_nms*_f32 needs for n boxes (suppression
 * bit-matrix uint64[n][ceil(n/64)], reference: iou3d_nms.cpp:99-103). */
size_t liso_iou3d_nms_workspace_bytes(int n);

/* replaces nms_gpu (iou3d_nms.cpp:90-136; kernel iou3d_nms_kernel.cu:267-311).
 * boxes must already be sorted by descending score.  The greedy sweep the
 * reference runs on the host after a blocking D2H copy (iou3d_nms.cpp:113-132)
 * runs on the device here:
 *   keep_dev[0 .. *num_out_dev) <- kept indices, ascending (int64, device)
 *   *num_out_dev                <- number kept (int32, device)
 * Both outputs are device memory; nothing is synchronised. */
int liso_iou3d_nms_f32(const float* boxes, int n, float thresh, int64_t* keep_dev, int* num_out_dev, void* workspace,
                       size_t workspace_bytes, void* stream);

/* replaces nms_normal_gpu (iou3d_nms.cpp:139-186; kernel iou3d_nms_kernel.cu:314-372). */
int liso_iou3d_nms_normal_f32(const float* boxes, int n, float thresh, int64_t* keep_dev, int* num_out_dev,
                              void* workspace, size_t workspace_bytes, void* stream);

/* replaces boxes_iou_bev_cpu (iou3d_cpu.cpp:232-252): host pointers, runs on the calling thread. */
int liso_iou3d_iou_bev_cpu_f32(const float* boxes_a, int n, const float* boxes_b, int m, float* out);

#ifdef __cplusplus
}
#endif
#endif /* LISO_IOU3D_H */
