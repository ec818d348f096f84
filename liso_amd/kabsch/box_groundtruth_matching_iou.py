"""Detection <-> ground-truth matching of the validation path.
Mirror of liso/kabsch/box_groundtruth_matching_iou.py:8-128; the greedy sweep runs on the device
(liso_match_greedy_f32, include/liso_tracking.h) on the IoU matrix produced by the HIP IoU kernels."""
import ctypes

import numpy as np
import torch

from liso_amd import _lib as L
from liso_amd.kabsch.shape_utils import Shape
from liso_amd.utils.nms_iou import box_iou_matrix


@torch.no_grad()
def greedy_match_iou_matrix(iou_matrix, pred_order, matching_threshold: float):
    """iou_matrix fp32 [n_gt,n_pred] (cuda), pred_order int64 [n_pred] most confident first -> device tensors
    (idx_gt, idx_pred, match_iou) padded to min(n_gt,n_pred), num_matches int32 [1], matched_pred_mask bool [n_pred],
    detected_gt_mask bool [n_gt].  No host synchronisation."""
    L.require_cuda(iou_matrix, pred_order)
    n_gt, n_pred = iou_matrix.shape
    iou = iou_matrix.float().t().contiguous()  # [n_pred, n_gt]: the walk reads one prediction's IoUs at a time
    order = pred_order.to(torch.int64).contiguous()
    dev, cap = iou.device, max(min(n_gt, n_pred), 1)
    idx_gt = torch.zeros(cap, dtype=torch.int64, device=dev)
    idx_pred = torch.zeros(cap, dtype=torch.int64, device=dev)
    miou = torch.zeros(cap, dtype=torch.float32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    pmask = torch.zeros(max(n_pred, 1), dtype=torch.uint8, device=dev)
    gmask = torch.zeros(max(n_gt, 1), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("match_greedy", lambda: L.lib().liso_match_greedy_f32(
            L.ptr(iou), 1, n_gt, n_gt, n_pred, L.ptr(order), float(matching_threshold), L.ptr(idx_gt), L.ptr(idx_pred), L.ptr(miou),
            L.ptr(num), L.ptr(pmask), L.ptr(gmask), L.stream_ptr())), "match_greedy")
    return idx_gt, idx_pred, miou, num, pmask[:n_pred].view(torch.bool), gmask[:n_gt].view(torch.bool)


def hungarian_match_iou_matrix(iou_matrix, matching_threshold: float):
    """reference :70-118, the "hungarian" branch: optimal assignment on the host (scipy.optimize.linear_sum_assignment, as the
    reference does it), the IoU matrix padded with -1 rows / columns to a square, non-finite entries -> -1, matches kept where
    IoU >= threshold.  iou_matrix: numpy [n_gt, n_pred] -> the reference's five arrays."""
    from scipy.optimize import linear_sum_assignment

    iou = np.array(iou_matrix, dtype=np.float64 if np.asarray(iou_matrix).dtype == np.float64 else np.asarray(iou_matrix).dtype, copy=True)
    n_true, n_pred = iou.shape
    MIN_IOU = -1.0
    if n_pred > n_true:
        iou = np.concatenate((iou, np.full((n_pred - n_true, n_pred), MIN_IOU)), axis=0)
    if n_true > n_pred:
        iou = np.concatenate((iou, np.full((n_true, n_true - n_pred), MIN_IOU)), axis=1)
    bad = ~np.isfinite(iou)
    if np.any(bad):
        iou[bad] = MIN_IOU
    idxs_true, matched_pred_idxs = linear_sum_assignment(iou, maximize=True)
    sel = matched_pred_idxs < n_pred
    idx_pred_actual, idx_gt_actual = matched_pred_idxs[sel], idxs_true[sel]
    ious_actual = iou[idx_gt_actual, idx_pred_actual]
    keep = ious_actual >= matching_threshold
    idxs_into_gt, idxs_into_preds, matching_dists = idx_gt_actual[keep], idx_pred_actual[keep], ious_actual[keep]
    det_gts_mask = np.zeros(n_true, dtype=bool)
    det_gts_mask[idxs_into_gt] = True
    matched_preds_mask = np.zeros(n_pred, dtype=bool)
    matched_preds_mask[idxs_into_preds] = True
    return idxs_into_gt, idxs_into_preds, matching_dists, matched_preds_mask, det_gts_mask


@torch.no_grad()
def match_boxes_by_descending_confidence_iou(non_batched_gt_boxes: Shape, non_batched_pred_boxes: Shape, matching_threshold: float,
                                             iou_mode: str = "iou_bev", matching_mode: str = "greedy"):
    """reference :8-128 -- same arguments; returns the reference's five numpy arrays (idxs_into_gt, idxs_into_preds,
    matching_dists, matched_preds_mask, det_gts_mask)."""
    assert iou_mode in ("iou_bev", "iou_3d"), iou_mode
    assert len(non_batched_gt_boxes.shape) == 1 and len(non_batched_pred_boxes.shape) == 1
    assert torch.all(non_batched_pred_boxes.valid), "need all valid predictions"
    assert torch.all(non_batched_gt_boxes.valid), "need all valid predictions"
    if matching_mode not in ("greedy", "hungarian"):
        raise NotImplementedError(matching_mode)
    n_pred, n_true = non_batched_pred_boxes.shape[0], non_batched_gt_boxes.shape[0]
    if matching_mode == "hungarian":  # IoU matrix from the HIP kernels, the assignment on the host like the reference
        iou = (np.zeros((n_true, n_pred)) if n_pred == 0 or n_true == 0
               else box_iou_matrix(non_batched_gt_boxes, non_batched_pred_boxes, iou_mode).cpu().numpy())
        return hungarian_match_iou_matrix(iou, matching_threshold)
    if n_pred == 0 or n_true == 0:  # reference :24-26: an empty IoU matrix, no pair to visit
        return (np.array([], dtype=np.int64), np.array([], dtype=np.int64), np.array([]), np.zeros(n_pred, dtype=bool),
                np.zeros(n_true, dtype=bool))
    iou = box_iou_matrix(non_batched_gt_boxes, non_batched_pred_boxes, iou_mode)
    order = torch.argsort(torch.squeeze(non_batched_pred_boxes.probs, dim=-1), descending=True)
    idx_gt, idx_pred, miou, num, pmask, gmask = greedy_match_iou_matrix(iou, order, matching_threshold)
    m = int(num.item())  # the function's only device->host sync; the results below are what the reference returns on the host
    return (idx_gt[:m].cpu().numpy(), idx_pred[:m].cpu().numpy(), miou[:m].cpu().numpy(), pmask.cpu().numpy(),
            gmask.cpu().numpy())
