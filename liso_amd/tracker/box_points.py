"""Host wrapper of liso_points_in_boxes_f32 (include/liso_tracking.h): per-box point counts, mean flow and the optional
[N,K] mask in one pass over the points.  Used by the mirrors of get_points_in_boxes_mask
(liso/datasets/torch_dataset_commons.py:1902-1935), Shape.get_points_in_box_bool_mask (liso/kabsch/shape_utils.py:488-538)
and propagate_boxes_forward_using_flow (liso/tracker/tracking.py:2168-2211)."""
import ctypes

import torch

from liso_amd import _lib as L

FP64_PRODUCT, FP32_PRODUCT = 0, 1  # liso_boxpts_cfg.precision


def dense_boxes(shape):
    """Shape [..,K] -> float32 [..,K,7] (x, y, z, dx, dy, dz, yaw).  No zeroing of invalid rows: the reference's inside
    test reads pos / dims / rot of every row as they are."""
    assert shape.pos.shape[-1] == 3 and shape.dims.shape[-1] == 3, (shape.pos.shape, shape.dims.shape)
    rot = shape.rot if shape.rot is not None and shape.rot.shape[-1] > 0 else torch.zeros_like(shape.pos[..., :1])
    return torch.cat([shape.pos, shape.dims, rot[..., :1]], dim=-1).float().contiguous()


@torch.no_grad()
def points_in_boxes(boxes7, points, point_valid=None, flow=None, want_mask=False, want_count=True, precision=FP64_PRODUCT,
                    dims_bloat=1.0):
    """boxes7 [B,K,7], points [B,N,>=3] (cuda, fp32) -> dict(count int32 [B,K], mean_flow fp32 [B,K,3] (if flow given),
    mask bool [B,N,K] (if want_mask))."""
    L.require_cuda(boxes7, points)
    assert boxes7.dim() == 3 and boxes7.shape[-1] == 7 and points.dim() == 3 and points.shape[0] == boxes7.shape[0]
    b, k, n = boxes7.shape[0], boxes7.shape[1], points.shape[1]
    boxes7, points = boxes7.float().contiguous(), points.float().contiguous()
    dev = points.device
    cfg = L.BoxPtsCfg(b, n, k, points.shape[-1], int(precision), float(dims_bloat))
    out = {}
    mask = torch.empty((b, n, k), dtype=torch.uint8, device=dev) if want_mask else None
    count = torch.empty((b, k), dtype=torch.int32, device=dev) if (want_count or flow is not None) else None
    mean, ws, ws_bytes, valid_u8, fl = None, None, 0, None, None
    if flow is not None:
        fl = flow.float().contiguous()
        assert fl.shape == (b, n, 3), fl.shape
        mean = torch.empty((b, k, 3), dtype=torch.float32, device=dev)
        ws_bytes = int(L.lib().liso_points_in_boxes_workspace_bytes(ctypes.byref(cfg)))
        ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=dev)
        if point_valid is not None:
            valid_u8 = point_valid.to(torch.uint8).contiguous()
            assert valid_u8.shape == (b, n), valid_u8.shape
    opt = lambda t: L.ptr(t) if t is not None else None  # noqa: E731
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("points_in_boxes", lambda: L.lib().liso_points_in_boxes_f32(
            ctypes.byref(cfg), L.ptr(boxes7), L.ptr(points), opt(valid_u8), opt(fl), opt(mask), opt(count), opt(mean), opt(ws),
            ws_bytes, L.stream_ptr())), "points_in_boxes")
    if mask is not None:
        out["mask"] = mask.view(torch.bool)  # 0/1 bytes
    if count is not None:
        out["count"] = count
    if mean is not None:
        out["mean_flow"] = mean
    return out
