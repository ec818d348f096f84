"""OpenPCDet-style wrappers, mirror of iou3d_nms/iou3d_nms_utils.py (same four functions)."""
import numpy as np
import torch

from liso_amd import iou3d_nms_cuda


def boxes_iou_bev(boxes_a, boxes_b):
    """reference :12-26"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans)
    return ans


def to_pcdet(boxes):
    """reference :28-32 -- swap dx/dy and re-reference the heading (note: fancy indexing copies, like the reference)."""
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]]
    boxes[:, -1] = -boxes[:, -1] - np.pi / 2
    return boxes


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """reference :34-72"""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    boxes_a, boxes_b = to_pcdet(boxes_a), to_pcdet(boxes_b)
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), bev)
    h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    o3d = bev * h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return o3d / torch.clamp(vol_a + vol_b - o3d, min=1e-6)


def _nms(fn, boxes, scores, thresh, pre_maxsize=None):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = fn(boxes, keep, thresh)
    return order[keep[:num_out].to(boxes.device)].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """reference :75-90"""
    return _nms(iou3d_nms_cuda.nms_gpu, boxes, scores, thresh, pre_maxsize)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """reference :92-106"""
    return _nms(iou3d_nms_cuda.nms_normal_gpu, boxes, scores, thresh)
