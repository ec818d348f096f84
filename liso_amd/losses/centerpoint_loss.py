"""CenterPoint losses.  Mirror of liso/losses/centerpoint_loss.py (vector / direct rotation; class_bins is not on the
hot path).  Same keyword-only signature and returned dict keys.

Boolean-mask indexing in the reference (x[mask]) forces a device->host sync per term; here every term is a masked
sum over the dense maps, which is the same arithmetic up to summation order and never leaves the device.
"""
from typing import Dict

import torch
import torch.nn.functional as F


def centerpoint_loss(*, loss_cfg: Dict, decoded_pred_box_maps: Dict, raw_activated_pred_box_maps: Dict, gt_maps: Dict,
                     gt_center_mask, rotation_loss_weights_map, box_prediction_cfg: Dict, ignore_region_is_true_mask):
    losses = {}
    num_pos = torch.clip(gt_center_mask.sum(), min=1.0)  # reference :25
    sel = gt_center_mask & ~ignore_region_is_true_mask
    self = sel[..., None].float()
    if "probs" in gt_maps:
        losses["loss/supervised/centermaps/probs"] = prob_heatmap_loss(
            loss_cfg, gt_center_mask, gt_maps["probs"], raw_activated_pred_box_maps["probs"], ignore_region_is_true_mask)
    if "rot" in gt_maps:
        if box_prediction_cfg.rotation_representation.method not in ("direct", "vector"):
            raise NotImplementedError(box_prediction_cfg.rotation_representation.method)
        # reference :37-60: weights = max(w,0.1) / max(sum,1) over the selected cells, L1 summed
        w = torch.clamp(rotation_loss_weights_map, min=0.1) * self
        w = w / torch.clamp(w.sum(), min=1.0)
        rot_loss = (torch.abs(raw_activated_pred_box_maps["rot"] - gt_maps["rot"]) * w).sum()
        losses["loss/supervised/centermaps/rot"] = 10 * rot_loss  # absent in the reference when no positive: then 0
    if "dims" in gt_maps:
        # reference :116-128: l1_loss(mean over K*3 elements).sum() / num_pos
        n_el = torch.clamp(self.sum() * gt_maps["dims"].shape[-1], min=1.0)
        d = (torch.abs(decoded_pred_box_maps["dims"] - gt_maps["dims"]) * self).sum() / n_el
        losses["loss/supervised/centermaps/dims"] = d / num_pos
    if "pos" in gt_maps:
        n_el = torch.clamp(self.sum() * gt_maps["pos"].shape[-1], min=1.0)
        p = (torch.abs(decoded_pred_box_maps["pos"] - gt_maps["pos"]) * self).sum() / n_el
        losses["loss/supervised/centermaps/pos"] = p / num_pos
    return losses


def prob_heatmap_loss(loss_cfg, gt_center_mask, groundtruth_probs, pred_logits, ignore_where_true_mask=None):
    """reference :139-162"""
    if loss_cfg.supervised.centermaps.confidence_target not in ("gaussian",):
        raise NotImplementedError(loss_cfg.supervised.centermaps.confidence_target)
    return compute_focal_loss(gt_center_mask, groundtruth_probs, pred_logits, 2.0, 0.5, ignore_where_true_mask)


def compute_focal_loss(gt_center_mask, groundtruth_probs, pred_logits, gamma, alpha, ignore_where_true_mask=None):
    """reference :165-200 -- CenterNet focal loss (alpha=0.5, gamma=2, beta=4)."""
    num_pos = torch.clip(gt_center_mask.sum(), min=1.0)
    probs_pos = torch.sigmoid(pred_logits)
    probs_neg = torch.sigmoid(-pred_logits)
    positive_loss = alpha * torch.pow(probs_neg, gamma) * F.logsigmoid(pred_logits)
    negative_loss = ((1 - alpha) * torch.pow(probs_pos, gamma) * torch.pow(1.0 - groundtruth_probs, 4.0)
                     * F.logsigmoid(-pred_logits))
    keep = torch.ones_like(gt_center_mask) if ignore_where_true_mask is None else ~ignore_where_true_mask
    pos_m = (gt_center_mask & keep)[..., None].float()
    neg_m = (~gt_center_mask & keep)[..., None].float()
    return -((positive_loss * pos_m).sum() + (negative_loss * neg_m).sum()) / num_pos


def rotation_vec_on_unit_circle(raw_activated_box_pred):
    """liso/kabsch/main_utils.py:51-58"""
    assert raw_activated_box_pred["rot"].shape[-1] == 2
    vector_len = torch.norm(raw_activated_box_pred["rot"], dim=-1)
    return F.mse_loss(input=vector_len, target=torch.ones_like(vector_len))
