"""Generates tests/golden/od_metrics_reference.npz from the reference's own liso/eval/od_metrics.py:
  ObjectDetectionMetrics(box_matching_criterion="iou_bev" | "iou_3d", use_slow_nuscenes_matching=True).update(...)  (:250-545)
on a few synthetic validation samples, then `get_conf_prec_rec` + `calc_ap` (:25-80) on the collected lists exactly as
`log_specific_pr_curve` (:814-857) calls them, and the true-positive error sums.  Matching runs through the reference's
`match_boxes_by_descending_confidence_iou` -> `box_iou_matrix` -> iou3d_nms_utils with the native module replaced as in
make_nms_iou_golden.py (numbers from oracle/_ref = the reference's iou3d_cpu.cpp compiled where it lies).
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_od_metrics_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_nms_iou_golden import _cpu_cuda, install_native_stub  # noqa: E402
from make_targets_golden import _Anything, import_with_stubs  # noqa: E402

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")


def scene(g, n_gt, n_pred, n_classes):
    """ground truth + predictions: ~60 % of the predictions are jittered copies of ground-truth boxes, the rest clutter"""
    def boxes(n):
        pos = np.concatenate([g.uniform(-45, 45, (n, 2)), g.uniform(-1.6, -0.4, (n, 1))], -1)
        dims = np.stack([g.uniform(2, 6, n), g.uniform(1, 2.5, n), g.uniform(1.2, 2.2, n)], -1)
        return pos, dims, g.uniform(-np.pi, np.pi, (n, 1))
    gp, gd, gr = boxes(n_gt)
    pp, pd, pr = boxes(n_pred)
    k = min(n_gt, int(0.6 * n_pred))
    sel = g.permutation(n_gt)[:k]
    pp[:k] = gp[sel] + g.normal(0, 0.25, (k, 3))
    pd[:k] = gd[sel] * g.uniform(0.85, 1.15, (k, 3))
    pr[:k] = gr[sel] + g.normal(0, 0.15, (k, 1))
    f = lambda a: a.astype(np.float32)  # noqa: E731
    gt = dict(pos=f(gp), dims=f(gd), rot=f(gr), probs=np.ones((n_gt, 1), np.float32), velo=f(g.uniform(0, 2.0, (n_gt, 1)) * (g.uniform(size=(n_gt, 1)) < 0.5)),
              class_id=g.integers(0, n_classes, (n_gt, 1)).astype(np.int64), valid=np.ones(n_gt, bool))
    pred = dict(pos=f(pp), dims=f(pd), rot=f(pr), probs=f(g.uniform(0.05, 1.0, (n_pred, 1))), velo=np.zeros((n_pred, 1), np.float32),
                class_id=g.integers(0, n_classes, (n_pred, 1)).astype(np.int64), valid=np.ones(n_pred, bool))
    return gt, pred


def main():
    install_native_stub()

    def _imp():
        import liso.eval.od_metrics as odm
        from liso.kabsch.shape_utils import Shape
        return odm, Shape

    odm, Shape = import_with_stubs(_imp)
    S = lambda d: Shape(**{k: torch.from_numpy(v) for k, v in d.items()})  # noqa: E731
    out = {}
    cases = {
        # tag: (criterion, class names, class idxs, bev filter, abs range, samples [(n_gt, n_pred)], seed)
        "bev": ("iou_bev", ("overall",), (0,), None, (None, None), [(12, 20), (30, 25), (5, 0), (0, 7), (18, 40)], 1),
        "i3d": ("iou_3d", ("overall", "car", "ped"), (0, 1, 2), (-40.0, -40.0, 40.0, 40.0), (2.0, 50.0), [(25, 30), (14, 9), (40, 60)], 2),
    }
    with _cpu_cuda(), torch.no_grad():
        for tag, (crit, names, idxs, bev, (rmin, rmax), samples, seed) in cases.items():
            g = np.random.default_rng(seed)
            m = odm.ObjectDetectionMetrics(moving_velocity_thresh=0.5, class_names=names, class_idxs=idxs, use_slow_nuscenes_matching=True,
                                           box_matching_criterion=crit, filter_detections_by_bev_area_min_max_m=bev,
                                           min_eval_range_m=rmin, max_eval_range_m=rmax)
            out[f"{tag}_n_samples"] = np.array(len(samples))
            for i, (n_gt, n_pred) in enumerate(samples):
                gt, pred = scene(g, n_gt, n_pred, 3)
                for k, v in gt.items():
                    out[f"{tag}_s{i}_gt_{k}"] = v
                for k, v in pred.items():
                    out[f"{tag}_s{i}_pred_{k}"] = v
                m.update(non_batched_gt_boxes=S(gt), non_batched_pred_boxes=S(pred), sample_token=str(i))
            for cn in names:
                for thr in m.matching_thresholds:
                    for cat in sorted(m.extra_categories):
                        lab = np.concatenate(m.per_class_per_thresh_per_category_gt_labels[cn][thr][cat])
                        sc = np.concatenate(m.per_class_per_thresh_per_category_scores[cn][thr][cat])
                        fn = np.concatenate(m.per_class_per_thresh_per_category_is_fn[cn][thr][cat])
                        conf, prec, rec = odm.get_conf_prec_rec(lab, sc, fn)
                        key = f"{tag}_{cn}_{thr}_{cat}"
                        out[key + "_labels"], out[key + "_scores"], out[key + "_is_fn"] = lab, sc, fn.astype(bool)
                        out[key + "_conf"], out[key + "_prec"], out[key + "_rec"] = conf, prec, rec
                        out[key + "_ap"] = np.array(odm.calc_ap(prec, min_recall=m.min_recall, min_precision=m.min_precision))
                        out[key + "_num_objs"] = np.array(m.per_class_per_thresh_label_stats[cn][thr][cat])
                    e = m.per_class_per_thresh_tp_errors_running_stats[cn][thr]
                    out[f"{tag}_{cn}_{thr}_tp_errors"] = np.array([e["ATE"], e["ASE"], e["AOE"], e["tps"]], np.float64)
            print(tag, {k: float(v) for k, v in out.items() if k.startswith(tag) and k.endswith("overall_ap")})
    # the curve functions on their own: ties in the scores, no detections, non-interpolated
    g = np.random.default_rng(3)
    for tag, n in {"c0": 200, "c1": 7, "c2": 0}.items():
        lab = g.uniform(size=n) < 0.6
        sc = np.round(g.uniform(size=n), 1 if tag == "c0" else 3)
        fn = lab & (g.uniform(size=n) < 0.3)
        sc = np.where(fn, -np.inf, sc)
        out[f"{tag}_labels"], out[f"{tag}_scores"], out[f"{tag}_is_fn"] = lab, sc, fn
        if n:
            for interp in (True, False):
                c, p, r = odm.get_conf_prec_rec(lab, sc, fn, use_interpolation=interp)
                out[f"{tag}_conf_{int(interp)}"], out[f"{tag}_prec_{int(interp)}"], out[f"{tag}_rec_{int(interp)}"] = c, p, r
            out[f"{tag}_ap"] = np.array(odm.calc_ap(odm.get_conf_prec_rec(lab, sc, fn)[1], 0.1, 0.1))
    # the sklearn-curve score clean-up and the Waymo-style AP
    for tag, sc in {"m0": np.array([0.9, -np.inf, 0.2, 0.5, -np.inf]), "m1": np.array([-np.inf, -np.inf]), "m2": np.array([0.3])}.items():
        out[f"{tag}_scores_in"] = sc
        out[f"{tag}_scores_clean"] = odm.map_scores_from_neg_infs_to_actual_min_score(sc)
    for tag, (pr, rc) in {"w0": (np.array([1.0, 0.0]), np.array([0.0, 1.0])),
                          "w1": (np.array([1.0, 0.9, 0.7, 0.65, 0.2]), np.array([0.0, 0.04, 0.31, 0.33, 0.9])),
                          "w2": (np.linspace(1, 0.5, 30), np.linspace(0, 0.9, 30))}.items():
        p2, r2, ap = odm.waymo_precisions_recalls_apscore(pr.copy(), rc.copy())
        out[f"{tag}_prec_in"], out[f"{tag}_rec_in"], out[f"{tag}_prec"], out[f"{tag}_rec"], out[f"{tag}_ap"] = pr, rc, p2, r2, np.array(ap)
    np.savez_compressed(os.path.join(HERE, "od_metrics_reference.npz"), **out)
    print("arrays:", len(out))


if __name__ == "__main__":
    main()
