"""CPU: pins oracle/iou3d_oracle.c against the reference (golden vectors made from the unmodified reference TU,
and the TU itself when oracle/_ref is built)."""
import glob
import os

import numpy as np
import pytest

from oracle import iou3d as O


def _goldens(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "iou3d_*.npz")))
    assert files, "no iou3d golden fixtures"
    return files


def test_oracle_matches_golden_bitwise(golden_dir):
    for f in _goldens(golden_dir):
        g = np.load(f)
        b = g["boxes_sorted"]
        iou = O.boxes_iou_bev(b, b)
        ov = O.boxes_overlap_bev(b, b)
        # same libm, no FMA: the restatement must reproduce the reference bit for bit
        assert np.array_equal(iou.view(np.uint32), g["iou"].view(np.uint32)), f
        assert np.array_equal(ov.view(np.uint32), g["overlap"].view(np.uint32)), f
        for t in (0.1, 0.3, 0.7):
            assert np.array_equal(O.nms(b, t), g[f"keep_{int(t*100):03d}"]), (f, t)


def test_oracle_matches_reference_build_when_present():
    if O.ref_boxes_iou_bev(np.zeros((1, 7), np.float32), np.zeros((1, 7), np.float32)) is None:
        pytest.skip("oracle/_ref not built on this machine")
    for seed, n, spread in [(21, 300, 50.0), (22, 200, 5.0), (23, 64, 1.0)]:
        a, _ = O.random_boxes(n, seed, spread)
        b, _ = O.random_boxes(n + 7, seed + 100, spread)
        assert np.array_equal(O.boxes_iou_bev(a, b).view(np.uint32), O.ref_boxes_iou_bev(a, b).view(np.uint32))
        assert np.array_equal(O.boxes_overlap_bev(a, b).view(np.uint32), O.ref_boxes_overlap_bev(a, b).view(np.uint32))


def test_known_answers():
    # identical boxes -> 1, disjoint -> 0, half-shifted axis-aligned -> 1/3
    a = np.array([[0, 0, 0, 4, 2, 1, 0.3]], np.float32)
    assert abs(O.boxes_iou_bev(a, a)[0, 0] - 1.0) < 1e-5
    far = np.array([[100, 0, 0, 4, 2, 1, 0.3]], np.float32)
    assert O.boxes_iou_bev(a, far)[0, 0] == 0.0
    a0 = np.array([[0, 0, 0, 4, 2, 1, 0.0]], np.float32)
    s = np.array([[2, 0, 0, 4, 2, 1, 0.0]], np.float32)
    assert abs(O.boxes_iou_bev(a0, s)[0, 0] - 1.0 / 3.0) < 1e-5
    assert abs(O.boxes_overlap_bev(a0, s)[0, 0] - 4.0) < 1e-5
    # empty inputs
    e = np.zeros((0, 7), np.float32)
    assert O.boxes_iou_bev(e, a).shape == (0, 1)
    assert len(O.nms(e, 0.1)) == 0


def test_nms_normal_and_greedy_properties():
    b, s = O.random_boxes(300, 9, 10.0)
    b = b[np.argsort(-s, kind="stable")]
    for fn in (O.nms, O.nms_normal):
        k = fn(b, 0.1)
        assert np.all(np.diff(k) > 0) and k[0] == 0  # ascending, best box always kept
    # kept set is an independent set of the suppression graph, every dropped box is hit by an earlier kept one
    iou = O.boxes_iou_bev(b, b)
    k = O.nms(b, 0.1)
    sub = np.triu(iou[np.ix_(k, k)], 1)
    assert (sub <= 0.1).all()
    dropped = np.setdiff1d(np.arange(len(b)), k)
    for d in dropped:
        assert (iou[k[k < d], d] > 0.1).any()
    assert np.array_equal(O.nms_from_iou(iou, 0.1), k)
