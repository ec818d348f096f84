"""Validation metrics: AP curves and true-positive errors of the detections.

Mirror of the numeric part of liso/eval/od_metrics.py: `calc_ap` (:25-39), `get_conf_prec_rec` (:42-80), `scale_iou` /
`angle_diff` / `abs_yaw_diff` (:83-113), and `ObjectDetectionMetrics` (:161-545: range / class filters, per-threshold matching,
the overall / moving / still book-keeping, ATE / ASE / AOE sums) with `compute()` returning the metric dictionary that the
reference's `log()` (:1106-1311) fills -- same keys (`<prefix>/<criterion>/<class>/<category>/AP@<thr><unit>`, ...), without its
matplotlib figures and the sklearn ROC / DET curves.

Where the work is: per sample the IoU matrix and the greedy matching run on the device (`box_iou_matrix` -> HIP rotated IoU,
`liso_match_greedy_f32`); the book-keeping is a handful of small host arrays per sample, as in the reference.  The precision /
recall sweep at the end is a sort + two prefix sums + two 101-point interpolations over all detections of the validation set.  It
stays on the HOST in float64 on purpose: the 101 recall levels are compared for equality with the steps of the recall curve
(tp / n_gt), and the quotient has to be the correctly rounded one the reference's numpy computes -- the device's float64 division
differs from it in the last bit on some entries (measured on MI355X, scripts/debug_od_metrics_device.py), which moves a level to the
other side of a step and changes AP by ~1e-4.
"""
from typing import Dict, Tuple

import numpy as np
import torch

from liso_amd.kabsch.box_groundtruth_matching_iou import match_boxes_by_descending_confidence_iou
from liso_amd.kabsch.shape_utils import Shape


def calc_ap(precisions, min_recall: float, min_precision: float) -> float:
    """nuScenes-style average precision over the 101-point curve (reference :25-39)"""
    assert 0 <= min_precision < 1 and 0 <= min_recall <= 1
    prec = torch.as_tensor(precisions, dtype=torch.float64).clone()
    if min_recall != 0.0:
        assert prec.numel() == 101, prec.numel()
    prec = prec[round(100 * min_recall) + 1:]  # recalls above min_recall only
    prec = (prec - min_precision).clamp_min(0.0)
    prec = torch.where(torch.isnan(prec), torch.full_like(prec, float("nan")), prec)
    return float(prec.mean()) / (1.0 - min_precision)


def _interp(x, xp, fp, right):
    """np.interp for increasing `xp` (ties allowed: like numpy, the LAST of equal xp wins on the right side of a jump)"""
    n = xp.numel()
    j = torch.searchsorted(xp, x, right=True)  # xp[j-1] <= x < xp[j]
    lo = (j - 1).clamp(0, n - 1)
    hi = j.clamp(0, n - 1)
    x0, x1, y0, y1 = xp[lo], xp[hi], fp[lo], fp[hi]
    slope = (y1 - y0) / (x1 - x0)
    y = torch.where(x1 > x0, slope * (x - x0) + y0, y0)
    y = torch.where(j == 0, fp[0].expand_as(y), y)       # left of the table: first value
    y = torch.where(x == xp[-1], fp[-1].expand_as(y), y)
    return torch.where(x > xp[-1], torch.full_like(y, right), y)


def get_conf_prec_rec(all_gt, all_scores, all_is_fn, use_interpolation=True):
    """confidence / precision / recall along the detections sorted by descending score (reference :42-80).  False negatives
    (`all_is_fn`) only count in the recall denominator.  Returns float64 tensors."""
    gt = torch.as_tensor(np.asarray(all_gt), dtype=torch.bool)
    sc = torch.as_tensor(np.asarray(all_scores), dtype=torch.float64)
    fn = torch.as_tensor(np.asarray(all_is_fn), dtype=torch.bool)
    # the visiting order among EQUAL scores shapes the curve between them: take numpy's own argsort, as the reference does
    order = torch.from_numpy(np.argsort(-np.asarray(all_scores, dtype=np.float64)))
    keep = order[~fn[order]]
    tp = torch.cumsum(gt[keep].double(), 0)
    fp = torch.cumsum((~gt[keep]).double(), 0)
    conf = sc[keep]
    prec = tp / (fp + tp)
    rec = tp / float(torch.count_nonzero(gt))
    if use_interpolation:
        # numpy's grid, not torch.linspace: the two differ in the last bit at some levels, and a recall level that coincides with a
        # step of the recall curve must compare equal to it
        rec_interp = torch.from_numpy(np.linspace(0, 1, 101))
        if prec.numel() > 0:
            prec = _interp(rec_interp, rec, prec, right=0.0)
            conf = _interp(rec_interp, rec, conf, right=0.0)
        else:
            prec = torch.full_like(rec_interp, float("nan"))
            conf = torch.full_like(rec_interp, float("nan"))
        rec = rec_interp
    return conf, prec, rec


def scale_iou(box_sizes_a: np.ndarray, box_sizes_b: np.ndarray):
    """IoU of two boxes aligned in position and heading (reference :83-98)"""
    assert box_sizes_a.shape == box_sizes_b.shape and box_sizes_a.shape[-1] in (2, 3), (box_sizes_a.shape, box_sizes_b.shape)
    inter = np.prod(np.minimum(box_sizes_a, box_sizes_b), axis=-1)
    union = np.prod(box_sizes_a, axis=-1) + np.prod(box_sizes_b, axis=-1) - inter
    return inter / np.maximum(union, 1e-6)


def angle_diff(gt_yaw, pred_yaw, period: float = 2 * np.pi):
    """reference :101-106"""
    diff = (gt_yaw - pred_yaw + period / 2) % period - period / 2
    return np.where(diff > np.pi, diff - (2 * np.pi), diff)


def abs_yaw_diff(gt_yaw, pred_yaw, period: float = 2 * np.pi):
    return np.abs(angle_diff(gt_yaw, pred_yaw, period=period))


class ObjectDetectionMetrics:
    CATEGORIES = ("overall", "moving", "still")

    def __init__(self, moving_velocity_thresh: float, eval_movable_classes_as_one: bool = True, class_names: Tuple[str] = ("overall",),
                 class_idxs: Tuple[int] = (0,), min_precision=0.1, min_recall=0.1, use_slow_nuscenes_matching=True,
                 box_matching_criterion="iou_bev", iou_matching_thresholds=(0.25, 0.3, 0.4, 0.5),
                 filter_detections_by_bev_area_min_max_m=None, min_eval_range_m=None, max_eval_range_m=None):
        if box_matching_criterion not in ("iou_3d", "iou_bev", "dist"):
            raise NotImplementedError(box_matching_criterion)
        if not use_slow_nuscenes_matching and box_matching_criterion != "dist":
            raise NotImplementedError("IoU criteria are matched greedily by descending confidence (reference :320-341)")
        self.use_slow_nuscenes_matching = use_slow_nuscenes_matching
        self.min_eval_range_m, self.max_eval_range_m = min_eval_range_m, max_eval_range_m
        self.eval_movable_classes_as_one = eval_movable_classes_as_one
        if class_names == ("overall",):
            class_idxs = (0,)
        assert len(class_names) == len(class_idxs), (class_names, class_idxs)
        self.class_idxs, self.class_names = class_idxs, class_names
        if box_matching_criterion == "dist":  # reference :186-190: centre distance in metres
            self.tp_metric_thresh, self.threshold_unit, self.matching_thresholds = 2.0, "m", (0.5, 1.0, 2.0, 4.0)
        else:
            self.tp_metric_thresh, self.threshold_unit = 0.5, box_matching_criterion
            self.matching_thresholds = tuple(iou_matching_thresholds)
        assert self.tp_metric_thresh in self.matching_thresholds
        self.box_matching_criterion = box_matching_criterion
        self.bev_range_min_xy_m = self.bev_range_max_xy_m = None
        if filter_detections_by_bev_area_min_max_m is not None:
            self.bev_range_min_xy_m = torch.tensor(filter_detections_by_bev_area_min_max_m[:2])
            self.bev_range_max_xy_m = torch.tensor(filter_detections_by_bev_area_min_max_m[2:])
        self.min_precision, self.min_recall = min_precision, min_recall
        self.moving_velocity_thresh = moving_velocity_thresh
        new = lambda leaf: {c: {t: leaf() for t in self.matching_thresholds} for c in self.class_names}  # noqa: E731
        per_cat = lambda make: (lambda: {cat: make() for cat in self.CATEGORIES})  # noqa: E731
        self.gt_labels, self.scores, self.is_fn = new(per_cat(list)), new(per_cat(list)), new(per_cat(list))
        self.label_stats = new(per_cat(int))
        self.tp_errors = new(lambda: {"AOE": 0.0, "ASE": 0.0, "ATE": 0.0, "tps": 0})

    # ---- filters (reference :125-158) ---------------------------------------------------------------------------------------
    def _in_bev_range(self, boxes):
        xy = boxes.pos[:, :2]
        ok = ((xy >= self.bev_range_min_xy_m[None].to(xy.device)) & (xy <= self.bev_range_max_xy_m[None].to(xy.device))).all(dim=-1)
        boxes.valid = boxes.valid & ok
        return boxes.drop_padding_boxes()

    def _in_abs_range(self, boxes):
        r = torch.linalg.norm(boxes.pos[:, :2], dim=-1)
        boxes.valid = boxes.valid & (self.min_eval_range_m <= r) & (r < self.max_eval_range_m)
        return boxes.drop_padding_boxes()

    @staticmethod
    def _of_class(boxes, class_idx):
        boxes.valid = boxes.valid & (class_idx == torch.squeeze(boxes.class_id, dim=-1))
        return boxes.drop_padding_boxes()

    # ---- accumulation (reference :250-545) ------------------------------------------------------------------------------------
    def update(self, *, non_batched_gt_boxes: Shape, non_batched_pred_boxes: Shape, sample_token: str = ""):
        gt, pred = non_batched_gt_boxes.clone(), non_batched_pred_boxes.clone()
        if self.bev_range_min_xy_m is not None:
            gt, pred = self._in_bev_range(gt), self._in_bev_range(pred)
        if self.max_eval_range_m is not None and self.min_eval_range_m is not None:
            gt, pred = self._in_abs_range(gt), self._in_abs_range(pred)
        for class_idx, class_name in zip(self.class_idxs, self.class_names):
            if class_name == "overall":
                c_gt, c_pred = gt, pred
            else:
                c_gt, c_pred = self._of_class(gt.clone(), class_idx), self._of_class(pred.clone(), class_idx)
            for thr in self.matching_thresholds:
                self._update_class_threshold(c_gt, c_pred, thr, class_name)

    def _update_class_threshold(self, gt, pred, thr, class_name):
        assert bool(gt.valid.all()) and bool(pred.valid.all()), "invalid objects not supported!"
        if self.box_matching_criterion == "dist":  # reference :310-325 (greedy, 2-D) / :343-355 (optimal assignment, 2-D)
            from liso_amd.kabsch.box_groundtruth_matching import match_bboxes, slow_greedy_match_boxes_by_desending_confidence_by_dist
            if self.use_slow_nuscenes_matching:
                idx_gt, idx_pred, _, pred_mask, gt_mask = slow_greedy_match_boxes_by_desending_confidence_by_dist(
                    gt.pos, pred.pos, non_batched_pred_confidence=torch.squeeze(pred.probs, dim=-1), matching_threshold=thr, match_in_nd=2)
            else:
                idx_gt, idx_pred, _, pred_mask, gt_mask = match_bboxes(gt.pos, pred.pos, DIST_MATCHING_THRESHOLD=thr, match_in_nd=2)
        else:
            idx_gt, idx_pred, _, pred_mask, gt_mask = match_boxes_by_descending_confidence_iou(
                gt, pred, matching_threshold=thr, iou_mode=self.box_matching_criterion, matching_mode="greedy")
        g, p = gt.numpy(), pred.numpy()
        logits = np.squeeze(p.probs, axis=-1)
        moving = np.linalg.norm(g.velo, axis=-1) > self.moving_velocity_thresh
        assert gt_mask.shape == moving.shape, (gt_mask.shape, moving.shape)
        for category, ignore in (("moving", ~moving), ("still", moving), ("overall", np.zeros_like(moving))):
            self._update_category(gt_mask, pred_mask, logits, idx_pred, idx_gt, ignore, thr, category, class_name)
        n_tp = int(np.count_nonzero(gt_mask))
        e = self.tp_errors[class_name][thr]
        e["tps"] += n_tp
        if n_tp > 0:
            assert len(idx_gt) == len(idx_pred) == n_tp
            e["ATE"] += np.linalg.norm(g.pos[idx_gt, :2] - p.pos[idx_pred, :2], axis=-1).sum()
            e["ASE"] += (1.0 - scale_iou(g.dims[idx_gt, ...], p.dims[idx_pred, ...])).sum()
            if g.rot is not None and p.rot is not None:
                e["AOE"] += abs_yaw_diff(np.squeeze(g.rot[idx_gt, ...], axis=-1), np.squeeze(p.rot[idx_pred, ...], axis=-1)).sum()

    def _update_category(self, gt_mask, pred_mask, logits, idx_pred, idx_gt, ignore_gt, thr, category, class_name):
        self.label_stats[class_name][thr][category] += int(np.count_nonzero(~ignore_gt))
        ignore_matched = ignore_gt[idx_gt]
        use_pred = np.ones_like(pred_mask)
        use_pred[idx_pred] = ~ignore_matched  # a prediction matched to an ignored ground-truth box is neither TP nor FP
        s_logits, s_pred_mask, s_gt_mask = logits[use_pred], pred_mask[use_pred], gt_mask[~ignore_gt]
        n_tp = int(np.count_nonzero(s_gt_mask))
        assert n_tp == np.count_nonzero(s_pred_mask), "mismatch"
        n_fn = len(s_gt_mask) - n_tp
        fp_scores = s_logits[~s_pred_mask]
        L, S, F = self.gt_labels[class_name][thr][category], self.scores[class_name][thr][category], self.is_fn[class_name][thr][category]
        L.append(np.zeros(fp_scores.shape[0], dtype=bool)); S.append(fp_scores); F.append(np.zeros(fp_scores.shape[0], dtype=bool))
        L.append(np.ones(n_fn, dtype=bool)); S.append(-np.inf * np.ones(n_fn)); F.append(np.ones(n_fn, dtype=bool))
        if n_tp > 0:
            tp_scores = logits[idx_pred[~ignore_matched]]
            L.append(np.ones(n_tp, dtype=bool)); S.append(tp_scores); F.append(np.zeros(n_tp, dtype=bool))

    # ---- the numbers `log()` reports (reference :1106-1311, :814-857, :654-700) --------------------------------------------------
    def collected(self, class_name, thr, category):
        cat = lambda lists, dt: np.concatenate(lists[class_name][thr][category]).astype(dt) if lists[class_name][thr][category] else np.zeros(0, dt)  # noqa: E731
        return cat(self.gt_labels, bool), cat(self.scores, np.float64), cat(self.is_fn, bool)

    def compute(self, writer_prefix: str = "") -> Dict[str, float]:
        out = {}
        with_crit = writer_prefix.rstrip("/") + f"/{self.box_matching_criterion}/"
        for class_name in self.class_names:
            pre = with_crit.rstrip("/") + "/" + class_name + "/"
            for thr in self.matching_thresholds:
                for category in self.CATEGORIES:
                    _, prec, _ = get_conf_prec_rec(*self.collected(class_name, thr, category))
                    ap = calc_ap(prec, min_recall=self.min_recall, min_precision=self.min_precision)
                    out[pre + f"{category}/AP@{thr:.1f}{self.threshold_unit}"] = ap
                    if category == "overall":
                        out[pre.rstrip("/") + f"/AP@{thr:.1f}{self.threshold_unit}"] = ap
                    out[pre + f"/{category}/{thr:.1f}{self.threshold_unit}/num_objs"] = self.label_stats[class_name][thr][category]
                e = self.tp_errors[class_name][thr]
                for k, v in e.items():
                    out[pre + f"/{thr:.1f}{self.threshold_unit}/{k}"] = v if k == "tps" else v / max(e["tps"], 1e-6)
        return out


    # ---- ROC / detection-error-tradeoff curves (reference :547-653, :924-1105): the numbers behind the reference's figures ------------
    def roc_curves(self, class_name: str = "overall", category: str = "overall", writer_prefix: str = ""):
        """per matching threshold: (fpr, tpr, confidence thresholds decreasing) of sklearn's roc_curve with the false negatives'
        -inf scores mapped below the smallest real score, without the final point, and the area under the curve (0 when only one
        label is present) -- what log_roc_curves plots and writes as `<prefix>/area_under_roc_curve@<thr><unit>`.
        -> (curves {thr: dict(fpr, tpr, thresholds, area)}, metrics {label: area})"""
        from sklearn.metrics import roc_auc_score, roc_curve

        curves, metrics = {}, {}
        for thr in self.matching_thresholds:
            all_gt, all_scores, _ = self.collected(class_name, thr, category)
            scores = map_scores_from_neg_infs_to_actual_min_score(all_scores)
            if scores.size == 0:
                continue
            fpr, tpr, conf = roc_curve(all_gt, scores)
            area = 0.0 if len(np.unique(all_gt)) <= 1 else float(roc_auc_score(all_gt, scores))
            metrics[writer_prefix.rstrip("/") + f"/area_under_roc_curve@{thr:.1f}{self.threshold_unit}"] = area
            curves[thr] = {"fpr": fpr[:-1], "tpr": tpr[:-1], "thresholds": conf[:-1], "area": area}
        return curves, metrics

    def det_tp_fp_curves(self, class_name: str = "overall", category: str = "overall"):
        """per matching threshold: false-positive / false-negative rates over decreasing confidence thresholds (sklearn det_curve) and
        the absolute numbers of false / true positives (sklearn's binary classification curve), with the reference's special cases for
        samples that are all true or all false positives (:947-975); every array without its final point (:989-1003)"""
        from sklearn.metrics import det_curve
        from sklearn.metrics._ranking import _binary_clf_curve

        out = {}
        for thr in self.matching_thresholds:
            all_gt, all_scores, _ = self.collected(class_name, thr, category)
            cats = np.unique(all_gt)
            if cats.size == 0:
                continue
            scores = map_scores_from_neg_infs_to_actual_min_score(all_scores)
            if len(cats) == 1:
                desc = np.sort(scores)[::-1]
                only_tp = bool(cats[0])
                fp_rate = np.zeros_like(scores) if only_tp else np.ones_like(scores)
                fn_rate = np.zeros_like(scores) if only_tp else np.ones_like(scores)
                thr_ratio, thr_abs = desc, np.copy(desc)
                fps = all_gt.size * (np.zeros_like(desc) if only_tp else np.ones_like(desc))
                tps = all_gt.size * (np.ones_like(desc) if only_tp else np.zeros_like(desc))
            else:
                fp_rate, fn_rate, thr_ratio = det_curve(all_gt, scores)
                res = _binary_clf_curve(all_gt, scores)
                fps, tps, thr_abs = res[0], res[1], res[2]
            out[thr] = {"fp_rate": fp_rate[:-1], "fn_rate": fn_rate[:-1], "thresholds_for_ratios": thr_ratio[:-1], "num_fps": fps[:-1],
                        "num_tps": tps[:-1], "thresholds_abs": thr_abs[:-1]}
        return out


def map_scores_from_neg_infs_to_actual_min_score(relevant_scores):
    """reference :1837-1859: the -inf scores of false negatives become a finite value just below the smallest real score (10 % of the
    score range below it; -666 when there is no real score), for the sklearn-style curves"""
    scores = np.asarray(relevant_scores)
    real = scores[scores != -np.inf]
    if real.size == 0:
        print("Warning, no valid scores found: Replacing with magic number -666")
        fill = -666
    else:
        fill = real.min() - 0.1 * (real.max() - real.min())
    return np.where(scores == -np.inf, fill, scores)


def waymo_precisions_recalls_apscore(precisions, recalls, max_recall_gap=0.05):
    """reference :1862-1906: Waymo-style AP -- wherever two consecutive operating points are more than `max_recall_gap` apart in recall,
    points are inserted every `max_recall_gap` with the (conservative) precision of the point after the gap; AP = trapezoid area"""
    precisions, recalls = np.asarray(precisions, np.float64), np.asarray(recalls, np.float64)
    eps = 1e-6
    for _ in range(1000):
        gaps = np.where((np.abs(np.diff(recalls)) - eps) > max_recall_gap)[0]
        if gaps.size == 0:
            break
        i = int(gaps[0])
        width = recalls[i + 1] - recalls[i]
        assert width > 0.0, width
        n_new = int(width / max_recall_gap) - 1
        new_r = np.linspace(recalls[i] + max_recall_gap, recalls[i + 1] - max_recall_gap, n_new)
        precisions = np.insert(precisions, np.repeat(i + 1, n_new), np.repeat(precisions[i + 1], n_new))
        recalls = np.insert(recalls, np.repeat(i + 1, n_new), new_r)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    return precisions, recalls, trapz(precisions, recalls)
