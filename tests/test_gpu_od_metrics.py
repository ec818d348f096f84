"""ObjectDetectionMetrics with the matching on the device against the fixture generated from the reference's class."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {"bev": ("iou_bev", ("overall",), (0,), None, (None, None)),
         "i3d": ("iou_3d", ("overall", "car", "ped"), (0, 1, 2), (-40.0, -40.0, 40.0, 40.0), (2.0, 50.0))}


@pytest.mark.parametrize("tag", ["bev", "i3d"])
def test_metrics_match_reference_fixture(golden_dir, tag):
    from liso_amd.eval.od_metrics import ObjectDetectionMetrics
    from liso_amd.kabsch.shape_utils import Shape

    g = np.load(f"{golden_dir}/od_metrics_reference.npz")
    crit, names, idxs, bev, (rmin, rmax) = CASES[tag]
    m = ObjectDetectionMetrics(moving_velocity_thresh=0.5, class_names=names, class_idxs=idxs, box_matching_criterion=crit,
                               filter_detections_by_bev_area_min_max_m=bev, min_eval_range_m=rmin, max_eval_range_m=rmax)
    for i in range(int(g[f"{tag}_n_samples"])):
        S = lambda side: Shape(**{k: torch.from_numpy(g[f"{tag}_s{i}_{side}_{k}"]).cuda()  # noqa: E731
                                  for k in ("pos", "dims", "rot", "probs", "velo", "class_id", "valid")})
        m.update(non_batched_gt_boxes=S("gt"), non_batched_pred_boxes=S("pred"), sample_token=str(i))
    res = m.compute("val")
    for cn in names:
        for thr in m.matching_thresholds:
            for cat in m.CATEGORIES:
                key = f"{tag}_{cn}_{thr}_{cat}"
                lab, sc, fn = m.collected(cn, thr, cat)
                assert np.array_equal(lab, g[key + "_labels"]) and np.array_equal(fn, g[key + "_is_fn"]), key
                assert np.array_equal(sc.astype(np.float32), g[key + "_scores"].astype(np.float32)), key
                got, want = res[f"val/{crit}/{cn}/{cat}/AP@{thr:.1f}{crit}"], float(g[key + "_ap"])
                assert (np.isnan(got) and np.isnan(want)) or abs(got - want) <= 1e-9, (key, got, want)
                assert res[f"val/{crit}/{cn}//{cat}/{thr:.1f}{crit}/num_objs"] == int(g[key + "_num_objs"])
            ate, ase, aoe, tps = g[f"{tag}_{cn}_{thr}_tp_errors"]
            e = m.tp_errors[cn][thr]
            assert e["tps"] == int(tps)
            assert abs(e["ATE"] - ate) <= 1e-4 * max(ate, 1) and abs(e["ASE"] - ase) <= 1e-4 * max(ase, 1) and abs(e["AOE"] - aoe) <= 1e-4 * max(aoe, 1)
    assert res[f"val/{crit}/overall/AP@0.5{crit}"] == res[f"val/{crit}/overall/overall/AP@0.5{crit}"] > 0.1
