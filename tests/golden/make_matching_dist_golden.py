"""Generates tests/golden/matching_dist_reference.npz from the reference's own centre-distance matching
(liso/kabsch/box_groundtruth_matching.py: slow_greedy_match_boxes_by_desending_confidence_by_dist :154-229, match_bboxes :95-151,
batched_match_bboxes :8-92) and ObjectDetectionMetrics(box_matching_criterion="dist") (liso/eval/od_metrics.py:161-545) with both
matching variants, on synthetic validation samples; the collected label / score lists, AP values and true-positive errors are what
tests/test_matching_dist.py and tests/test_gpu_od_metrics.py pin the build to.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_matching_dist_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_nms_iou_golden import _cpu_cuda, install_native_stub  # noqa: E402
from make_od_metrics_golden import scene  # noqa: E402
from make_targets_golden import _Anything, import_with_stubs  # noqa: E402

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")


def main():
    install_native_stub()

    def _imp():
        import liso.eval.od_metrics as odm
        import liso.kabsch.box_groundtruth_matching as bgm
        from liso.kabsch.shape_utils import Shape
        return odm, bgm, Shape

    odm, bgm, Shape = import_with_stubs(_imp)
    S = lambda d: Shape(**{k: torch.from_numpy(v) for k, v in d.items()})  # noqa: E731
    out = {}
    g = np.random.default_rng(11)
    # ---- the matchers alone ------------------------------------------------------------------------------------------------------
    cases = [(12, 20, 2.0), (30, 25, 1.0), (5, 9, 4.0), (40, 60, 0.5), (3, 3, 2.0)]
    out["n_cases"] = np.array(len(cases))
    for i, (n_gt, n_pred, thr) in enumerate(cases):
        gt, pred = scene(g, n_gt, n_pred, 3)
        if i == 2:  # ties: two predictions at exactly the same distance from one ground-truth box, equal confidences elsewhere
            pred["pos"][1] = gt["pos"][0] + np.array([0.5, 0.0, 0.0], np.float32)
            pred["pos"][2] = gt["pos"][0] - np.array([0.5, 0.0, 0.0], np.float32)
        out[f"m{i}_gt_pos"], out[f"m{i}_pred_pos"], out[f"m{i}_conf"], out[f"m{i}_thr"] = gt["pos"], pred["pos"], pred["probs"][:, 0], np.array(thr)
        with torch.no_grad():
            r = bgm.slow_greedy_match_boxes_by_desending_confidence_by_dist(
                torch.from_numpy(gt["pos"]), torch.from_numpy(pred["pos"]), torch.from_numpy(pred["probs"][:, 0]), thr, match_in_nd=2)
            h = bgm.match_bboxes(torch.from_numpy(gt["pos"]), torch.from_numpy(pred["pos"]), DIST_MATCHING_THRESHOLD=thr, match_in_nd=2)
        for tag, res in (("greedy", r), ("hung", h)):
            for name, v in zip(("idx_gt", "idx_pred", "dists", "pred_mask", "gt_mask"), res):
                out[f"m{i}_{tag}_{name}"] = np.asarray(v)
    # batched optimal assignment on a padded batch
    gts, preds = zip(*[scene(g, 10, 14, 3) for _ in range(3)])
    bg = {k: np.stack([d[k] for d in gts]) for k in gts[0]}
    bp = {k: np.stack([d[k] for d in preds]) for k in preds[0]}
    bg["valid"][1, 7:] = False
    bp["valid"][2, 10:] = False
    for k, v in bg.items():
        out[f"b_gt_{k}"] = v
    for k, v in bp.items():
        out[f"b_pred_{k}"] = v
    with torch.no_grad():
        res = bgm.batched_match_bboxes(S(bg), S(bp), MAX_DIST_PADDING_VALUE=1000.0, DIST_MATCHING_THRESHOLD=2.0)
    for name, v in zip(("idx_gt", "idx_pred", "dists", "pred_mask", "gt_mask"), res):
        out[f"b_{name}"] = np.asarray(v)
    # ---- the metrics class with the distance criterion -----------------------------------------------------------------------------
    samples = [(12, 20), (30, 25), (5, 0), (0, 7), (18, 40)]
    with _cpu_cuda(), torch.no_grad():
        for tag, slow in (("dslow", True), ("dfast", False)):
            gg = np.random.default_rng(5)
            m = odm.ObjectDetectionMetrics(moving_velocity_thresh=0.5, class_names=("overall", "car"), class_idxs=(0, 1),
                                           use_slow_nuscenes_matching=slow, box_matching_criterion="dist")
            out[f"{tag}_n_samples"] = np.array(len(samples))
            for i, (n_gt, n_pred) in enumerate(samples):
                gt, pred = scene(gg, n_gt, n_pred, 3)
                for k, v in gt.items():
                    out[f"{tag}_s{i}_gt_{k}"] = v
                for k, v in pred.items():
                    out[f"{tag}_s{i}_pred_{k}"] = v
                m.update(non_batched_gt_boxes=S(gt), non_batched_pred_boxes=S(pred), sample_token=str(i))
            out[f"{tag}_thresholds"] = np.array(m.matching_thresholds)
            for cn in ("overall", "car"):
                for thr in m.matching_thresholds:
                    for cat in sorted(m.extra_categories):
                        key = f"{tag}_{cn}_{thr}_{cat}"
                        lab = np.concatenate(m.per_class_per_thresh_per_category_gt_labels[cn][thr][cat])
                        sc = np.concatenate(m.per_class_per_thresh_per_category_scores[cn][thr][cat])
                        fn = np.concatenate(m.per_class_per_thresh_per_category_is_fn[cn][thr][cat])
                        out[key + "_labels"], out[key + "_scores"], out[key + "_is_fn"] = lab, sc, fn
                        _, prec, _ = odm.get_conf_prec_rec(lab, sc, fn)
                        out[key + "_ap"] = np.array(odm.calc_ap(prec, min_recall=m.min_recall, min_precision=m.min_precision))
                        out[key + "_num"] = np.array(m.per_class_per_thresh_label_stats[cn][thr][cat])
                    e = m.per_class_per_thresh_tp_errors_running_stats[cn][thr]
                    out[f"{tag}_{cn}_{thr}_tp_errors"] = np.array([e["ATE"], e["ASE"], e["AOE"], e["tps"]], np.float64)
            # the numbers behind the ROC / DET figures (:547-600, :924-1003) for the overall class: same sklearn calls on the reference's lists
            from sklearn.metrics import det_curve, roc_auc_score, roc_curve
            for thr in m.matching_thresholds:
                lab = np.concatenate(m.per_class_per_thresh_per_category_gt_labels["overall"][thr]["overall"])
                sc = odm.map_scores_from_neg_infs_to_actual_min_score(np.concatenate(m.per_class_per_thresh_per_category_scores["overall"][thr]["overall"]))
                fpr, tpr, conf = roc_curve(lab, sc)
                out[f"{tag}_roc_{thr}_fpr"], out[f"{tag}_roc_{thr}_tpr"], out[f"{tag}_roc_{thr}_conf"] = fpr[:-1], tpr[:-1], conf[:-1]
                out[f"{tag}_roc_{thr}_area"] = np.array(roc_auc_score(lab, sc))
                fp_rate, fn_rate, t = det_curve(lab, sc)
                out[f"{tag}_det_{thr}_fp"], out[f"{tag}_det_{thr}_fn"], out[f"{tag}_det_{thr}_t"] = fp_rate[:-1], fn_rate[:-1], t[:-1]
    path = os.path.join(HERE, "matching_dist_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
