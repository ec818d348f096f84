"""Centre-distance matching (liso_amd/kabsch/box_groundtruth_matching.py) against the fixture written by the reference's own
functions (tests/golden/make_matching_dist_golden.py): host paths here (numpy greedy branch, optimal assignment), the device greedy
walk and the metrics class with the `dist` criterion in the GPU tests below."""
import numpy as np
import pytest
import torch

NAMES = ("idx_gt", "idx_pred", "dists", "pred_mask", "gt_mask")


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(f"{golden_dir}/matching_dist_reference.npz")


def _same(res, g, prefix, atol=5e-3):
    # (distances: torch.cdist switches to its matrix-multiplication formula above 25 rows -- a^2 + b^2 - 2ab in fp32, ~1e-3 m of
    # cancellation error at 45 m coordinates -- which the reference's numbers carry; indices and masks must be identical)
    for name, v in zip(NAMES, res):
        want = g[f"{prefix}_{name}"]
        if name == "dists":
            assert np.allclose(np.asarray(v, np.float64), want, atol=atol), (prefix, name)
        else:
            assert np.array_equal(np.asarray(v), want), (prefix, name, v, want)


def test_optimal_assignment_and_numpy_greedy_match_reference(g):
    from liso_amd.kabsch.box_groundtruth_matching import match_bboxes, slow_greedy_match_boxes_by_desending_confidence_by_dist

    for i in range(int(g["n_cases"])):
        gt, pred, conf, thr = g[f"m{i}_gt_pos"], g[f"m{i}_pred_pos"], g[f"m{i}_conf"], float(g[f"m{i}_thr"])
        _same(match_bboxes(torch.from_numpy(gt), torch.from_numpy(pred), DIST_MATCHING_THRESHOLD=thr, match_in_nd=2), g, f"m{i}_hung")
        # numpy branch of the greedy matcher (the reference's uses all position columns there): pass the 2-D positions
        res = slow_greedy_match_boxes_by_desending_confidence_by_dist(gt[:, :2], pred[:, :2], conf, thr)
        # (np.argsort(...)[::-1] orders ties differently from torch.argsort(descending=True): compare where confidences are distinct)
        if len(np.unique(conf)) == len(conf):
            _same(res, g, f"m{i}_greedy")


def test_batched_optimal_assignment_matches_reference(g):
    from liso_amd.kabsch.box_groundtruth_matching import batched_match_bboxes
    from liso_amd.kabsch.shape_utils import Shape

    S = lambda side: Shape(**{k: torch.from_numpy(g[f"b_{side}_{k}"]) for k in ("pos", "dims", "rot", "probs", "velo", "class_id", "valid")})  # noqa: E731
    res = batched_match_bboxes(S("gt"), S("pred"), MAX_DIST_PADDING_VALUE=1000.0, DIST_MATCHING_THRESHOLD=2.0)
    _same(res, g, "b")


@pytest.mark.gpu
def test_device_greedy_distance_matching_matches_reference(g):
    from liso_amd.kabsch.box_groundtruth_matching import slow_greedy_match_boxes_by_desending_confidence_by_dist

    for i in range(int(g["n_cases"])):
        gt, pred, conf, thr = g[f"m{i}_gt_pos"], g[f"m{i}_pred_pos"], g[f"m{i}_conf"], float(g[f"m{i}_thr"])
        res = slow_greedy_match_boxes_by_desending_confidence_by_dist(
            torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda(), torch.from_numpy(conf).cuda(), thr, match_in_nd=2)
        _same(res, g, f"m{i}_greedy")
    # host tensors (the sequence tracker's per-frame boxes live on the host, as in the reference): the reference's own host calls
    for i in range(int(g["n_cases"])):
        gt, pred, conf, thr = g[f"m{i}_gt_pos"], g[f"m{i}_pred_pos"], g[f"m{i}_conf"], float(g[f"m{i}_thr"])
        _same(slow_greedy_match_boxes_by_desending_confidence_by_dist(torch.from_numpy(gt), torch.from_numpy(pred), torch.from_numpy(conf), thr,
                                                                      match_in_nd=2), g, f"m{i}_greedy")


@pytest.mark.gpu
@pytest.mark.parametrize("tag,slow", [("dslow", True), ("dfast", False)])
def test_metrics_with_distance_criterion_match_reference(g, tag, slow):
    from liso_amd.eval.od_metrics import ObjectDetectionMetrics
    from liso_amd.kabsch.shape_utils import Shape

    m = ObjectDetectionMetrics(moving_velocity_thresh=0.5, class_names=("overall", "car"), class_idxs=(0, 1), use_slow_nuscenes_matching=slow,
                               box_matching_criterion="dist")
    assert m.matching_thresholds == tuple(g[f"{tag}_thresholds"]) and m.threshold_unit == "m" and m.tp_metric_thresh == 2.0
    for i in range(int(g[f"{tag}_n_samples"])):
        S = lambda side: Shape(**{k: torch.from_numpy(g[f"{tag}_s{i}_{side}_{k}"]).cuda()  # noqa: E731
                                  for k in ("pos", "dims", "rot", "probs", "velo", "class_id", "valid")})
        m.update(non_batched_gt_boxes=S("gt"), non_batched_pred_boxes=S("pred"), sample_token=str(i))
    res = m.compute("val")
    for cn in ("overall", "car"):
        for thr in m.matching_thresholds:
            for cat in m.CATEGORIES:
                key = f"{tag}_{cn}_{thr}_{cat}"
                lab, sc, fn = m.collected(cn, thr, cat)
                assert np.array_equal(lab, g[key + "_labels"]) and np.array_equal(fn, g[key + "_is_fn"]), key
                assert np.array_equal(sc.astype(np.float32), g[key + "_scores"].astype(np.float32)), key
                got, want = res[f"val/dist/{cn}/{cat}/AP@{thr:.1f}m"], float(g[key + "_ap"])
                assert (np.isnan(got) and np.isnan(want)) or abs(got - want) <= 1e-9, (key, got, want)
                assert res[f"val/dist/{cn}//{cat}/{thr:.1f}m/num_objs"] == int(g[key + "_num"])
            ate, ase, aoe, tps = g[f"{tag}_{cn}_{thr}_tp_errors"]
            e = m.tp_errors[cn][thr]
            assert e["tps"] == int(tps) and abs(e["ATE"] - ate) <= 1e-4 * max(ate, 1) and abs(e["ASE"] - ase) <= 1e-4 * max(ase, 1)
    # the numbers behind the reference's ROC / detection-error-tradeoff figures
    curves, metrics = m.roc_curves("overall", "overall", "val")
    det = m.det_tp_fp_curves("overall", "overall")
    for thr in m.matching_thresholds:
        assert np.allclose(curves[thr]["fpr"], g[f"{tag}_roc_{thr}_fpr"]) and np.allclose(curves[thr]["tpr"], g[f"{tag}_roc_{thr}_tpr"])
        assert np.allclose(curves[thr]["thresholds"], g[f"{tag}_roc_{thr}_conf"], rtol=1e-6)
        assert abs(metrics[f"val/area_under_roc_curve@{thr:.1f}m"] - float(g[f"{tag}_roc_{thr}_area"])) <= 1e-9
        assert np.allclose(det[thr]["fp_rate"], g[f"{tag}_det_{thr}_fp"]) and np.allclose(det[thr]["fn_rate"], g[f"{tag}_det_{thr}_fn"])
        assert len(det[thr]["num_fps"]) == len(det[thr]["num_tps"]) == len(det[thr]["thresholds_abs"])
