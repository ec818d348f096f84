"""Centre-distance matching of detections and ground truth.  Mirror of
liso/kabsch/box_groundtruth_matching.py: `slow_greedy_match_boxes_by_desending_confidence_by_dist` (:154-229, nuScenes-style greedy
matching by descending confidence), `match_bboxes` (:95-151) and `batched_match_bboxes` (:8-92, optimal assignment) -- same names,
arguments and return values.

Device tensors: the distance matrix is torch.cdist on the device (what the reference calls); the greedy walk runs on the device in
liso_match_greedy_f32 (include/liso_tracking.h) on the NEGATED distances -- "largest score above -threshold, first ground-truth index
on ties" is exactly "smallest distance below the threshold, first index on ties" (:198-217).  The optimal assignment stays
scipy.optimize.linear_sum_assignment on the host, as in the reference.  numpy inputs take the reference's numpy branch on the host."""
from typing import Union

import numpy as np
import torch

from liso_amd import _lib as L


def _greedy_host(dist_matrix, sortind, matching_threshold):
    """reference :193-229 on a host distance matrix [n_true, n_pred]"""
    n_true, n_pred = dist_matrix.shape
    matched_preds_mask, det_gts_mask = np.zeros(n_pred, dtype=bool), np.zeros(n_true, dtype=bool)
    idxs_into_gt, idxs_into_preds, matching_dists = [], [], []
    taken = np.zeros(n_true, dtype=bool)
    for pred_idx in sortind:
        min_dist, match_gt_idx = np.inf, None
        for gt_idx in range(n_true):
            if not taken[gt_idx] and dist_matrix[gt_idx, pred_idx] < min_dist:
                min_dist, match_gt_idx = dist_matrix[gt_idx, pred_idx], gt_idx
        if min_dist < matching_threshold:
            taken[match_gt_idx] = True
            idxs_into_gt.append(match_gt_idx), idxs_into_preds.append(pred_idx), matching_dists.append(min_dist)
            matched_preds_mask[pred_idx], det_gts_mask[match_gt_idx] = True, True
    return (np.array(idxs_into_gt, dtype=np.int64), np.array(idxs_into_preds, dtype=np.int64), np.array(matching_dists),
            matched_preds_mask, det_gts_mask)


@torch.no_grad()
def slow_greedy_match_boxes_by_desending_confidence_by_dist(non_batched_gt_boxes_pos: Union[torch.Tensor, np.ndarray],
                                                            non_batched_pred_boxes_pos: Union[torch.Tensor, np.ndarray],
                                                            non_batched_pred_confidence: Union[torch.Tensor, np.ndarray],
                                                            matching_threshold: float, match_in_nd=3):
    """reference :154-229 -> (idxs_into_gt, idxs_into_preds, matching_dists, matched_preds_mask, det_gts_mask) as numpy arrays"""
    assert len(non_batched_gt_boxes_pos.shape) == 2, non_batched_gt_boxes_pos.shape
    assert len(non_batched_pred_boxes_pos.shape) == 2, non_batched_pred_boxes_pos.shape
    assert len(non_batched_pred_confidence.shape) == 1, non_batched_pred_confidence.shape
    n_pred, n_true = non_batched_pred_boxes_pos.shape[0], non_batched_gt_boxes_pos.shape[0]
    assert non_batched_pred_confidence.shape[0] == n_pred, (non_batched_pred_confidence.shape[0], n_pred)
    if not torch.is_tensor(non_batched_gt_boxes_pos):  # numpy branch (reference :186-191; note: all position columns, as there)
        from scipy.spatial import distance_matrix

        dist = distance_matrix(non_batched_gt_boxes_pos.astype(np.float32), non_batched_pred_boxes_pos.astype(np.float32))
        return _greedy_host(dist, np.argsort(non_batched_pred_confidence)[::-1], matching_threshold)
    if not non_batched_gt_boxes_pos.is_cuda:
        # host tensors (the sequence tracker keeps its few boxes per frame on the host, as the reference does,
        # liso/tracker/global_box_tracker.py:56-57): the same calls on the host -- torch.argsort's order of equal confidences is the
        # host's, which decides the association order when several tracks have confidence 1
        if n_pred == 0 or n_true == 0:
            return (np.array([], dtype=np.int64), np.array([], dtype=np.int64), np.array([]), np.zeros(n_pred, dtype=bool),
                    np.zeros(n_true, dtype=bool))
        dist = torch.cdist(non_batched_gt_boxes_pos[..., :match_in_nd].to(torch.float32),
                           non_batched_pred_boxes_pos[..., :match_in_nd].to(torch.float32)).numpy()
        return _greedy_host(dist, torch.argsort(non_batched_pred_confidence, descending=True).numpy(), matching_threshold)
    L.require_cuda(non_batched_gt_boxes_pos, non_batched_pred_boxes_pos, non_batched_pred_confidence)
    if n_pred == 0 or n_true == 0:
        return (np.array([], dtype=np.int64), np.array([], dtype=np.int64), np.array([]), np.zeros(n_pred, dtype=bool),
                np.zeros(n_true, dtype=bool))
    from liso_amd.kabsch.box_groundtruth_matching_iou import greedy_match_iou_matrix

    dist = torch.cdist(non_batched_gt_boxes_pos[..., :match_in_nd].to(torch.float32),
                       non_batched_pred_boxes_pos[..., :match_in_nd].to(torch.float32))          # [n_gt, n_pred]
    order = torch.argsort(non_batched_pred_confidence, descending=True)
    idx_gt, idx_pred, neg, num, pmask, gmask = greedy_match_iou_matrix(-dist, order, -float(matching_threshold))
    m = int(num.item())  # the only device->host synchronisation; the reference returns host arrays
    return (idx_gt[:m].cpu().numpy(), idx_pred[:m].cpu().numpy(), (-neg[:m]).cpu().numpy(), pmask.cpu().numpy(), gmask.cpu().numpy())


def _pad_square(dist_matrix, n_true, n_pred, pad):
    if n_pred > n_true:  # dummy rows (ground truths)
        dist_matrix = np.concatenate((dist_matrix, np.full(dist_matrix.shape[:-2] + (n_pred - n_true, n_pred), pad)), axis=-2)
    if n_true > n_pred:  # dummy columns (predictions)
        dist_matrix = np.concatenate((dist_matrix, np.full(dist_matrix.shape[:-2] + (n_true, n_true - n_pred), pad)), axis=-1)
    return dist_matrix


@torch.no_grad()
def match_bboxes(gt_pos, pred_pos, DIST_MATCHING_THRESHOLD=15.0, match_in_nd=3):
    """reference :95-151: optimal assignment on the centre distances (padded to a square with 1000 m), pairs closer than the
    threshold are matches"""
    from scipy.optimize import linear_sum_assignment

    assert len(gt_pos.shape) == 2, gt_pos.shape
    assert len(pred_pos.shape) == 2, pred_pos.shape
    n_pred, n_true = pred_pos.shape[0], gt_pos.shape[0]
    MAX_DIST = 1000.0
    dist_matrix = torch.cdist(gt_pos[..., :match_in_nd].to(torch.float32), pred_pos[..., :match_in_nd].to(torch.float32)).cpu().numpy()
    dist_matrix = _pad_square(dist_matrix, n_true, n_pred, MAX_DIST)
    idxs_true, matched_pred_idxs = linear_sum_assignment(dist_matrix)
    sel_pred = matched_pred_idxs < n_pred
    idx_pred_actual, idx_gt_actual = matched_pred_idxs[sel_pred], idxs_true[sel_pred]
    dists_actual = dist_matrix[idx_gt_actual, idx_pred_actual]
    keep = dists_actual < DIST_MATCHING_THRESHOLD
    idxs_gt, idxs_pred, matching_dists = idx_gt_actual[keep], idx_pred_actual[keep], dists_actual[keep]
    detected_gts_mask = np.zeros(n_true, dtype=bool)
    detected_gts_mask[idxs_gt] = True
    matched_preds_mask = np.zeros(n_pred, dtype=bool)
    matched_preds_mask[idxs_pred] = True
    assert np.count_nonzero(detected_gts_mask) == np.count_nonzero(matched_preds_mask)
    return idxs_gt, idxs_pred, matching_dists, matched_preds_mask, detected_gts_mask


@torch.no_grad()
def batched_match_bboxes(groundtruth_bboxes, predicted_bboxes, MAX_DIST_PADDING_VALUE, DIST_MATCHING_THRESHOLD):
    """reference :8-92: the same per sample of a padded batch of `Shape`s; invalid boxes are pushed to MAX_DIST_PADDING_VALUE"""
    from scipy.optimize import linear_sum_assignment

    pred_pos, gt_pos = predicted_bboxes.pos.detach(), groundtruth_bboxes.pos.detach()
    dist_mat = torch.cdist(gt_pos.to(torch.float32), pred_pos)
    dist_mat = torch.where(groundtruth_bboxes.valid[:, :, None], dist_mat, MAX_DIST_PADDING_VALUE)
    dist_mat = torch.where(predicted_bboxes.valid[:, None, :], dist_mat, MAX_DIST_PADDING_VALUE)
    dist_matrix = dist_mat.cpu().numpy()
    bs, n_pred, _ = pred_pos.shape
    _, n_true, _ = gt_pos.shape
    dist_matrix = _pad_square(dist_matrix, n_true, n_pred, MAX_DIST_PADDING_VALUE)
    idxs_true, matched_pred_idxs = zip(*[linear_sum_assignment(dist_matrix[i]) for i in range(bs)])
    idxs_true, matched_pred_idxs = np.stack(idxs_true, axis=0), np.stack(matched_pred_idxs, axis=0)
    max_pad = max(n_pred, n_true)
    batch_idxs = np.tile(np.arange(0, bs, 1, dtype=np.int64)[..., None], (1, max_pad))
    idxs_true = np.stack([batch_idxs, idxs_true], axis=-1)
    matched_pred_idxs = np.stack([batch_idxs, matched_pred_idxs], axis=-1)
    sel_pred = matched_pred_idxs[..., 1] < n_pred
    idx_pred_actual, idx_gt_actual = matched_pred_idxs[sel_pred], idxs_true[sel_pred]
    assert (idx_gt_actual[..., 0] == idx_pred_actual[..., 0]).all(), "cross-batch match occured"
    dists_actual = dist_matrix[idx_gt_actual[..., 0], idx_gt_actual[..., 1], idx_pred_actual[..., 1]]
    matched_preds_mask = dists_actual < DIST_MATCHING_THRESHOLD
    idxs_gt, idxs_pred, matching_dists = idx_gt_actual[matched_preds_mask], idx_pred_actual[matched_preds_mask], dists_actual[matched_preds_mask]
    detected_gts_mask = np.zeros((bs, n_true), dtype=bool)
    detected_gts_mask[idxs_gt[..., 0], idxs_gt[..., 1]] = True
    return idxs_gt, idxs_pred, matching_dists, matched_preds_mask, detected_gts_mask
