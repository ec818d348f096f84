"""Drop-in for the reference's pybind module ``iou3d_nms_cuda``.

Same five names, argument order, ownership rules and return values as
iou3d_nms/src/iou3d_nms_api.cpp:11-17 (prototypes iou3d_nms/src/iou3d_nms.h:9-12,
iou3d_nms/src/iou3d_cpu.h:9), backed by the gfx950 kernels behind the C ABI of
include/liso_iou3d.h.  Differences, all intentional:

* bad inputs raise ``LisoHipError`` instead of printing and calling ``exit(-1)``
  (iou3d_nms.cpp:14-38);
* kernels run on the *current* PyTorch HIP stream, not the legacy default stream;
* ``nms_gpu`` runs the greedy sweep on the device; only the final ``keep`` list is
  copied to the caller's CPU ``LongTensor`` (the reference copies the whole bit-matrix,
  iou3d_nms.cpp:108-110).  ``nms_gpu_device`` skips even that copy.
"""
import torch

from . import _lib as L


def _check_boxes(name, t, device=True):
    if device:
        L.require_cuda(t)
    elif t.is_cuda:
        raise L.LisoHipError(f"{name} must be a CPU tensor")
    if t.dtype != torch.float32:
        raise L.LisoHipError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise L.LisoHipError(f"{name} must be contiguous")  # CHECK_CONTIGUOUS, iou3d_nms.cpp:20-25
    if t.dim() != 2 or t.shape[1] != 7:
        raise L.LisoHipError(f"{name} must be [N,7], got {tuple(t.shape)}")


def _check_out(name, t, n, m, device=True):
    if device:
        L.require_cuda(t)
    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n * m:
        raise L.LisoHipError(f"{name} must be a contiguous float32 tensor with {n}x{m} elements")


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:49-68.  ans_overlap[N,M] (caller-allocated, device) <- BEV overlap area."""
    _check_boxes("boxes_a", boxes_a)
    _check_boxes("boxes_b", boxes_b)
    n, m = boxes_a.shape[0], boxes_b.shape[0]
    _check_out("ans_overlap", ans_overlap, n, m)
    with torch.cuda.device(boxes_a.device):
        L.check(L.lib().liso_iou3d_overlap_bev_f32(L.ptr(boxes_a), n, L.ptr(boxes_b), m, L.ptr(ans_overlap),
                                                   L.stream_ptr()), "boxes_overlap_bev_gpu")
    return 1


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """iou3d_nms.cpp:70-88.  ans_iou[N,M] (caller-allocated, device) <- rotated BEV IoU."""
    _check_boxes("boxes_a", boxes_a)
    _check_boxes("boxes_b", boxes_b)
    n, m = boxes_a.shape[0], boxes_b.shape[0]
    _check_out("ans_iou", ans_iou, n, m)
    with torch.cuda.device(boxes_a.device):
        L.check(L.lib().liso_iou3d_iou_bev_f32(L.ptr(boxes_a), n, L.ptr(boxes_b), m, L.ptr(ans_iou), L.stream_ptr()),
                "boxes_iou_bev_gpu")
    return 1


def _nms_device(boxes, thresh, normal):
    _check_boxes("boxes", boxes)
    n = boxes.shape[0]
    dev = boxes.device
    keep_dev = torch.empty(n, dtype=torch.int64, device=dev)
    num_dev = torch.empty(1, dtype=torch.int32, device=dev)
    lib = L.lib()
    ws_bytes = lib.liso_iou3d_nms_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
    fn = lib.liso_iou3d_nms_normal_f32 if normal else lib.liso_iou3d_nms_f32
    with torch.cuda.device(dev):
        L.check(fn(L.ptr(boxes), n, float(thresh), L.ptr(keep_dev), L.ptr(num_dev), L.ptr(ws), ws_bytes,
                   L.stream_ptr()), "nms_normal_gpu" if normal else "nms_gpu")
    return keep_dev, num_dev


def nms_gpu_device(boxes, thresh):
    """Device-resident NMS: returns (keep_dev int64[N], num_dev int32[1]); first num entries valid. No sync."""
    return _nms_device(boxes, thresh, normal=False)


def nms_normal_gpu_device(boxes, thresh):
    return _nms_device(boxes, thresh, normal=True)


def _nms_host_api(boxes, keep, thresh, normal):
    if keep.is_cuda or keep.dtype != torch.int64 or not keep.is_contiguous():
        raise L.LisoHipError("keep must be a contiguous CPU LongTensor (iou3d_nms.cpp:94,100)")
    n = boxes.shape[0]
    if keep.numel() < n:
        raise L.LisoHipError("keep is shorter than boxes")
    keep_dev, num_dev = _nms_device(boxes, thresh, normal)
    num = int(num_dev.item())  # the reference API returns a host int: one sync is inherent
    if num:
        keep[:num].copy_(keep_dev[:num])
    return num


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:90-136.  boxes sorted by descending score; keep: CPU LongTensor[N]; returns #kept."""
    return _nms_host_api(boxes, keep, nms_overlap_thresh, normal=False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:139-186 (axis-aligned IoU, heading ignored)."""
    return _nms_host_api(boxes, keep, nms_overlap_thresh, normal=True)


def boxes_iou_bev_cpu(boxes_a_tensor, boxes_b_tensor, ans_iou_tensor):
    """iou3d_cpu.cpp:232-252: CPU tensors in, CPU tensor out (the reference's own CPU entry point)."""
    _check_boxes("boxes_a", boxes_a_tensor, device=False)
    _check_boxes("boxes_b", boxes_b_tensor, device=False)
    n, m = boxes_a_tensor.shape[0], boxes_b_tensor.shape[0]
    _check_out("ans_iou", ans_iou_tensor, n, m, device=False)
    L.check(L.lib().liso_iou3d_iou_bev_cpu_f32(L.ptr(boxes_a_tensor), n, L.ptr(boxes_b_tensor), m,
                                               L.ptr(ans_iou_tensor)), "boxes_iou_bev_cpu")
    return 1
