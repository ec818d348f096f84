"""NMS / IoU python boundary, mirror of liso/utils/nms_iou.py (same names, arguments and results), backed by the
gfx950 kernels through liso_amd.iou3d_nms_cuda.  shapely_nms (CPU polygon fallback, reference :210-227) is out of
scope: liso_amd has no CPU fallbacks."""
from typing import List

import torch

from liso_amd import iou3d_nms_cuda
from liso_amd.kabsch.shape_utils import Shape


def hard_limit_detections(non_batched_pred_boxes, max_num_centerpoint_preds):
    """reference :10-20 -- keep the top-k most confident boxes."""
    top = torch.argsort(torch.squeeze(non_batched_pred_boxes.probs, dim=-1), dim=0, descending=True)[
        : min(max_num_centerpoint_preds, non_batched_pred_boxes.pos.shape[0])]
    mask = torch.zeros_like(non_batched_pred_boxes.valid)
    mask[top] = True
    non_batched_pred_boxes.valid = mask
    return non_batched_pred_boxes.drop_padding_boxes()


def perform_nms_on_shapes(pred_visu_boxes: Shape, max_num_boxes: int, overlap_threshold: float, use_cuda=True,
                          pre_nms_max_num_boxes=-1):
    """reference :23-66 -- per-sample NMS over a batched Shape."""
    if not use_cuda:
        raise NotImplementedError("shapely CPU NMS is out of scope (SURVEY.md 2.1); use_cuda=True only")
    out = []
    boxes = pred_visu_boxes.clone()
    for b in range(boxes.pos.shape[0]):
        s = boxes[b].drop_padding_boxes()
        if pre_nms_max_num_boxes > 0 and s.shape[0] > pre_nms_max_num_boxes:
            keep_mask = torch.zeros_like(s.valid)
            desc = torch.argsort(s.probs, dim=0, descending=True)
            keep_mask[desc[:pre_nms_max_num_boxes]] = True
            s.valid = s.valid & keep_mask
            s = s.drop_padding_boxes()
        idxs = iou_based_nms(s, overlap_threshold=overlap_threshold)
        out.append(hard_limit_detections(s[idxs], max_num_boxes))
    return Shape.from_list_of_shapes(out)


@torch.no_grad()
def iou_based_nms(objects: Shape, overlap_threshold: float, pre_nms_max_boxes: int = None,
                  post_nms_max_boxes: int = None) -> List[int]:
    """reference :78-99"""
    assert len(objects.probs.shape) == 2, objects.probs.shape
    assert objects.probs.shape[-1] == 1, objects.probs.shape
    assert objects.valid.all(), "can't handle padding boxes"
    boxes_conv = convert_shapes_to_dense_3d(objects.clone())
    assert boxes_conv.shape[-1] == 7, boxes_conv.shape
    return rotate_nms_pcdet(boxes_conv, torch.squeeze(objects.probs, dim=-1), overlap_threshold,
                            pre_maxsize=pre_nms_max_boxes, post_max_size=post_nms_max_boxes)


def boxes_iou_bev(boxes_a, boxes_b):
    """reference :102-121 -- (N,7),(M,7) -> (N,M)"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans)
    return ans


@torch.no_grad()
def box_iou_matrix(boxes_a: Shape, boxes_b: Shape, iou_mode: str = "iou_bev"):
    """reference :124-207 -- BEV IoU, or 3-D IoU = BEV overlap x height overlap / union volume."""
    for s in (boxes_a, boxes_b):
        assert len(s.probs.shape) == 2 and s.probs.shape[-1] == 1, s.probs.shape
        assert s.valid.all(), "can't handle padding boxes"
    ca = convert_shapes_to_dense_3d(boxes_a.clone())
    cb = convert_shapes_to_dense_3d(boxes_b.clone())
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    if na == 0 or nb == 0:
        return torch.zeros((na, nb), device=ca.device)
    out = torch.zeros((na, nb), dtype=torch.float32, device=ca.device)
    if iou_mode == "iou_bev":
        iou3d_nms_cuda.boxes_iou_bev_gpu(ca.float().contiguous(), cb.float().contiguous(), out)
        return out
    if iou_mode != "iou_3d":
        raise NotImplementedError(iou_mode)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(ca.float().contiguous(), cb.float().contiguous(), out)
    lo_a = boxes_a.pos[:, 2] - 0.5 * boxes_a.dims[:, 2]
    lo_b = boxes_b.pos[:, 2] - 0.5 * boxes_b.dims[:, 2]
    hi_a = boxes_a.pos[:, 2] + 0.5 * boxes_a.dims[:, 2]
    hi_b = boxes_b.pos[:, 2] + 0.5 * boxes_b.dims[:, 2]
    h = torch.min(hi_a[:, None], hi_b[None, :]) - torch.max(lo_a[:, None], lo_b[None, :])
    inter = torch.where(h > 0.0, out * h, torch.zeros_like(out))
    union = torch.prod(boxes_a.dims, dim=-1)[:, None] + torch.prod(boxes_b.dims, dim=-1)[None, :] - inter
    return inter / torch.clip(union, min=torch.finfo(torch.float32).eps)


def convert_shapes_to_dense_3d(boxes: Shape):
    """reference :230-242 -- Shape -> [N,7] (x,y,z,dx,dy,dz,heading); invalid rows zeroed."""
    dense = torch.cat([pad_attr_to_3d_if_necessary(boxes.pos, 0.0), pad_attr_to_3d_if_necessary(boxes.dims, 1.0),
                       boxes.rot], dim=-1)
    return torch.where(boxes.valid[..., None], dense, torch.zeros_like(dense))


def pad_attr_to_3d_if_necessary(pos_or_dims, padding_value: float):
    """reference :245-254"""
    if pos_or_dims.shape[-1] == 2:
        return torch.cat([pos_or_dims, padding_value * torch.ones_like(pos_or_dims[..., [0]])], dim=-1)
    if pos_or_dims.shape[-1] == 3:
        return pos_or_dims
    raise NotImplementedError("can't handle shape", pos_or_dims.shape)


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """reference :257-282 -- sort by score, pre-top-k, rotated NMS, post-top-k; returns indices into `boxes`.
    The kept list never leaves the device (the reference bounces it through a CPU LongTensor, :270-277); the
    only host sync is reading the count, which the variable-length return value makes inherent."""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    if len(boxes) == 0:
        selected = order[:0].contiguous()
    else:
        keep_dev, num_dev = iou3d_nms_cuda.nms_gpu_device(boxes.float().contiguous(), thresh)
        selected = order[keep_dev[: int(num_dev.item())]].contiguous()
    if post_max_size is not None:
        selected = selected[:post_max_size]
    return selected


@torch.no_grad()
def perform_nms_on_shapes_padded(boxes: Shape, max_num_boxes: int, overlap_threshold: float, pre_nms_max_num_boxes=-1,
                                 return_target_arrays=False):
    """`perform_nms_on_shapes` (reference :23-66) without leaving the device: the same survivors -- per sample: boxes by
    descending confidence (stable), at most `pre_nms_max_num_boxes` into rotated NMS, the first `max_num_boxes` survivors
    kept -- returned in that order inside a padded Shape [B,K] whose `valid` flags mark them (the reference compacts each
    sample on the host: 4+ device->host syncs per sample).  Invalid input slots never suppress anything; the slots that are not
    kept come back with the padding values of `Shape.set_padding_val_to(0.0)` (shape_utils.py:439-462).
    Three launches per sample around the NMS kernels (include/liso_box_mining.h): no sort / scan library call, no host read.
    `return_target_arrays`: also return (pos fp32 [B,K,3], dims fp32 clamped to >= 1e-3, rot fp32 [B,K], valid uint8 [B,K]),
    the arrays the CenterPoint target renderer takes."""
    from liso_amd.networks.flow_cluster_detector import mining_ops as MO

    B, K = boxes.valid.shape
    if K == 0:
        return (boxes, None) if return_target_arrays else boxes
    want = {"pos": torch.float32, "dims": torch.float64, "rot": torch.float64, "probs": torch.float64, "velo": torch.float64,
            "class_id": torch.int32, "difficulty": torch.int32}
    orig = {k: getattr(boxes, k).dtype for k in want}
    arrays, fresh = {}, []  # fresh: (new buffer, caller's tensor) pairs copied by ONE launch (_lib.multi_copy)
    for k, dt in want.items():
        v = getattr(boxes, k)
        if k in ("pos", "dims"):
            v = pad_attr_to_3d_if_necessary(v, 0.0 if k == "pos" else 1.0)
        if k == "velo" and v.shape[-1] != 1:
            raise NotImplementedError("padded NMS: velo with more than one component")
        v = v.to(dt)
        if v.data_ptr() == getattr(boxes, k).data_ptr() and v.is_contiguous() and v.is_cuda:  # (permuted in place: never the caller's memory)
            arrays[k] = torch.empty_like(v)
            fresh.append((arrays[k], v))
        else:
            arrays[k] = v.clone() if v.data_ptr() == getattr(boxes, k).data_ptr() else v.contiguous()
        if not arrays[k].is_contiguous():
            arrays[k] = arrays[k].contiguous()
    arrays["valid"] = boxes.valid.to(torch.uint8).contiguous()
    if arrays["valid"].data_ptr() == boxes.valid.data_ptr():
        arrays["valid"] = arrays["valid"].clone()
    if fresh:
        from liso_amd import _lib as L

        L.multi_copy(fresh)
    t_arrays = MO.nms_select(arrays, max_num_boxes, overlap_threshold, pre_nms_max_num_boxes)
    out = Shape(pos=arrays["pos"].to(orig["pos"]), dims=arrays["dims"].to(orig["dims"]), rot=arrays["rot"].to(orig["rot"]),
                probs=arrays["probs"].to(orig["probs"]), velo=arrays["velo"].to(orig["velo"]), valid=arrays["valid"].view(torch.bool),
                class_id=arrays["class_id"].to(orig["class_id"]), difficulty=arrays["difficulty"].to(orig["difficulty"]))
    return (out, t_arrays) if return_target_arrays else out
