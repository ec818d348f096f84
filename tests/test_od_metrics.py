"""Precision / recall curves and AP of the validation path (liso_amd/eval/od_metrics.py) against the fixture generated from the
reference's own `get_conf_prec_rec` / `calc_ap` (tests/golden/make_od_metrics_golden.py).  Host part: runs without a GPU."""
import numpy as np
import pytest

from liso_amd.eval.od_metrics import abs_yaw_diff, calc_ap, get_conf_prec_rec, scale_iou


def test_curves_match_reference_incl_score_ties_and_empty(golden_dir):
    g = np.load(f"{golden_dir}/od_metrics_reference.npz")
    for tag in ("c0", "c1"):
        for interp in (1, 0):
            c, p, r = get_conf_prec_rec(g[f"{tag}_labels"], g[f"{tag}_scores"], g[f"{tag}_is_fn"], use_interpolation=bool(interp))
            for got, key in ((c, "conf"), (p, "prec"), (r, "rec")):
                want = g[f"{tag}_{key}_{interp}"]
                assert got.shape == want.shape
                assert np.allclose(got.numpy(), want, rtol=0, atol=1e-12, equal_nan=True), (tag, key, interp)
        assert abs(calc_ap(get_conf_prec_rec(g[f"{tag}_labels"], g[f"{tag}_scores"], g[f"{tag}_is_fn"])[1], 0.1, 0.1) - float(g[f"{tag}_ap"])) <= 1e-12
    c, p, r = get_conf_prec_rec(g["c2_labels"], g["c2_scores"], g["c2_is_fn"])  # nothing collected: NaN curves, like the reference
    assert p.shape == (101,) and bool(np.isnan(p.numpy()).all()) and bool(np.isnan(c.numpy()).all())


def test_ap_of_every_collected_list_matches_reference(golden_dir):
    g = np.load(f"{golden_dir}/od_metrics_reference.npz")
    keys = [k[:-len("_labels")] for k in g.files if k.endswith("_labels") and k[0] in "bi"]
    assert len(keys) == 4 * 3 + 3 * 4 * 3
    for key in keys:
        c, p, r = get_conf_prec_rec(g[key + "_labels"], g[key + "_scores"], g[key + "_is_fn"])
        assert np.allclose(p.numpy(), g[key + "_prec"], rtol=0, atol=1e-12, equal_nan=True), key
        assert np.allclose(c.numpy(), g[key + "_conf"], rtol=0, atol=1e-12, equal_nan=True), key
        ap, want = calc_ap(p, 0.1, 0.1), float(g[key + "_ap"])
        assert (np.isnan(ap) and np.isnan(want)) or abs(ap - want) <= 1e-12, key


def test_tp_error_helpers():
    assert np.allclose(scale_iou(np.array([[2.0, 1.0, 1.0]]), np.array([[1.0, 2.0, 1.0]])), 1.0 / 3.0)
    assert np.allclose(abs_yaw_diff(np.array([3.1, -3.1, 0.2]), np.array([-3.1, 3.1, -0.1])), [2 * np.pi - 6.2, 2 * np.pi - 6.2, 0.3])
    with pytest.raises(AssertionError):
        calc_ap(np.zeros(50), 0.1, 0.1)


def test_score_cleanup_and_waymo_ap_match_reference(golden_dir):
    from liso_amd.eval.od_metrics import map_scores_from_neg_infs_to_actual_min_score, waymo_precisions_recalls_apscore

    g = np.load(f"{golden_dir}/od_metrics_reference.npz")
    for tag in ("m0", "m1", "m2"):
        assert np.array_equal(map_scores_from_neg_infs_to_actual_min_score(g[f"{tag}_scores_in"]), g[f"{tag}_scores_clean"]), tag
    for tag in ("w0", "w1", "w2"):
        p, r, ap = waymo_precisions_recalls_apscore(g[f"{tag}_prec_in"], g[f"{tag}_rec_in"])
        assert np.allclose(p, g[f"{tag}_prec"], atol=1e-12) and np.allclose(r, g[f"{tag}_rec"], atol=1e-12), tag
        assert abs(ap - float(g[f"{tag}_ap"])) <= 1e-12, tag
    assert abs(float(g["w0_ap"]) - 0.025) < 1e-9  # p(0) = 1, p(1) = 0: the trapezoid over the inserted points gives 0.025 (the comment in the reference says 0.05)
