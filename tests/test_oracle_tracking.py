"""The numpy restatement of the tracking / validation inner loops (oracle/tracking.py) against fixtures produced by the
reference's own python (tests/golden/make_tracking_golden.py)."""
import os

import numpy as np
import pytest

from oracle import tracking as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tracking_reference.npz"))


def unpack(bits, n):
    return np.unpackbits(bits, axis=0)[:n].astype(bool)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_point_masks_match_reference(tag):
    pos, dims, rot, pts = G[f"{tag}_pos"], G[f"{tag}_dims"], G[f"{tag}_rot"], G[f"{tag}_pts"]
    n = pts.shape[0]
    assert np.array_equal(O.points_in_boxes_mask(pos, dims, rot, pts), unpack(G[f"{tag}_mask64"], n))
    # the fp32 product rounds in BLAS order in the reference: a point within one fp32 ulp of a face may flip
    for key, bloat in (("mask32", 1.0), ("mask32_bloat", 1.25)):
        got, ref = O.points_in_box_bool_mask(pos, dims, rot, pts, bloat), unpack(G[f"{tag}_{key}"], n)
        assert (got != ref).sum() <= 1, (got != ref).sum()
        assert ref.sum() > 0 or tag == "c"


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_flow_propagation_matches_reference(tag):
    a = {k: G[f"{tag}_{k}"] for k in ("pos", "dims", "rot", "pts", "valid", "flow", "odom")}
    fg, bg, warped, st1 = O.propagate_boxes_forward_using_flow(a["pos"], a["dims"], a["rot"], a["pts"], a["valid"], a["flow"],
                                                               a["odom"])
    np.testing.assert_allclose(fg, G[f"{tag}_fg"][0], atol=2e-6)  # fp32 sums over <= 20k points
    np.testing.assert_allclose(bg, G[f"{tag}_bg"][0, 0], atol=1e-12)
    np.testing.assert_allclose(warped, G[f"{tag}_warped"], atol=5e-6)
    np.testing.assert_allclose(st1, G[f"{tag}_st1"][0], atol=5e-6)
    assert np.abs(fg[:, :3, 3]).max() > 0.1 or tag == "c"


@pytest.mark.parametrize("tag", ["m0", "m1", "m2", "m3", "m4"])
@pytest.mark.parametrize("thr", [0.3, 0.5])
def test_greedy_matching_matches_reference(tag, thr):
    ig, ip, d, pm, gm = O.match_greedy(G[f"{tag}_iou"], G[f"{tag}_conf"], thr)
    assert np.array_equal(ig, G[f"{tag}_{thr}_idx_gt"]) and np.array_equal(ip, G[f"{tag}_{thr}_idx_pred"])
    assert np.array_equal(d, G[f"{tag}_{thr}_dists"])
    assert np.array_equal(pm, G[f"{tag}_{thr}_pred_mask"]) and np.array_equal(gm, G[f"{tag}_{thr}_gt_mask"])
