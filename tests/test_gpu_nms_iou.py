"""the python boundary liso.utils.nms_iou / iou3d_nms.iou3d_nms_utils on the GPU against a fixture
produced by the REFERENCE's own python (tests/golden/make_nms_iou_golden.py; its native module backed by the compiled
reference CPU IoU of oracle/_ref).  IoU values <= 1e-5 abs (fp32 geometry, see tests/test_gpu_iou3d.py), index lists
identical."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "nms_iou_reference.npz"))


def _shape(fx, prefix):
    from liso_amd.kabsch.shape_utils import Shape

    return Shape(**{k: torch.from_numpy(fx[f"{prefix}_{k}"]).cuda() for k in ("pos", "dims", "rot", "probs", "valid")})


@pytest.mark.parametrize("tag", ["m3", "m2", "me"])
def test_box_iou_matrix_and_dense_conversion(fx, tag):
    from liso_amd.utils.nms_iou import box_iou_matrix, convert_shapes_to_dense_3d

    a, b = _shape(fx, f"{tag}_a"), _shape(fx, f"{tag}_b")
    assert np.array_equal(convert_shapes_to_dense_3d(a.clone()).cpu().numpy(), fx[f"{tag}_dense_a"])
    got = box_iou_matrix(a, b, iou_mode="iou_bev").cpu().numpy()
    assert got.shape == fx[f"{tag}_iou_bev"].shape
    if got.size:
        assert np.abs(got - fx[f"{tag}_iou_bev"]).max() <= 1e-5
        assert np.array_equal(got == 0.0, fx[f"{tag}_iou_bev"] == 0.0)
    if f"{tag}_iou_3d" in fx.files:
        got3 = box_iou_matrix(a, b, iou_mode="iou_3d").cpu().numpy()
        assert got3.shape == fx[f"{tag}_iou_3d"].shape
        if got3.size:
            assert np.abs(got3 - fx[f"{tag}_iou_3d"]).max() <= 1e-5
            assert (got3 > 0).sum() > 10  # the fixture really has overlapping pairs with height overlap


@pytest.mark.parametrize("tag", ["n1", "n2", "n3", "n4"])
def test_iou_based_nms_with_pre_post_topk(fx, tag):
    from liso_amd.utils.nms_iou import iou_based_nms

    pre, post, thr = fx[f"{tag}_cfg"]
    keep = iou_based_nms(_shape(fx, tag), float(thr), pre_nms_max_boxes=None if pre < 0 else int(pre),
                         post_nms_max_boxes=None if post < 0 else int(post))
    assert np.array_equal(keep.cpu().numpy(), fx[f"{tag}_keep"])


def test_perform_nms_on_shapes_batched(fx):
    from liso_amd.utils.nms_iou import perform_nms_on_shapes, perform_nms_on_shapes_padded

    boxes = _shape(fx, "p")
    res = perform_nms_on_shapes(boxes.clone(), max_num_boxes=30, overlap_threshold=0.1, pre_nms_max_num_boxes=100)
    for k in ("pos", "dims", "rot", "probs", "valid"):
        assert np.array_equal(getattr(res, k).cpu().numpy(), fx[f"p_out_{k}"], equal_nan=True), k
    # the sync-free variant keeps the same boxes per sample, in the same order
    pad = perform_nms_on_shapes_padded(boxes.clone(), max_num_boxes=30, overlap_threshold=0.1, pre_nms_max_num_boxes=100)
    for b in range(3):
        ref_valid = fx["p_out_valid"][b]
        got = pad[b].drop_padding_boxes()
        assert got.shape[0] == int(ref_valid.sum())
        assert np.array_equal(got.pos.cpu().numpy(), fx["p_out_pos"][b][ref_valid])


def test_openpcdet_wrappers(fx):
    from liso_amd.iou3d_nms import iou3d_nms_utils as U

    a, b = torch.from_numpy(fx["u_a"]).cuda(), torch.from_numpy(fx["u_b"]).cuda()
    sc = torch.from_numpy(fx["u_scores"]).cuda()
    assert np.abs(U.to_pcdet(a.clone()).cpu().numpy() - fx["u_to_pcdet"]).max() <= 1e-6
    assert np.abs(U.boxes_iou_bev(a, b).cpu().numpy() - fx["u_iou_bev"]).max() <= 1e-5
    got3 = U.boxes_iou3d_gpu(a.clone(), b.clone()).cpu().numpy()
    assert np.abs(got3 - fx["u_iou3d"]).max() <= 1e-5 and (fx["u_iou3d"] > 0).sum() > 10
    assert np.array_equal(U.nms_gpu(a, sc, 0.1, pre_maxsize=60)[0].cpu().numpy(), fx["u_nms"])
    assert np.array_equal(U.nms_normal_gpu(a, sc, 0.1)[0].cpu().numpy(), fx["u_nms_normal"])
