"""The "hungarian" branch of match_boxes_by_descending_confidence_iou (reference :70-118) against a fixture written by the
reference's own function (tests/golden/make_matching_hungarian_golden.py; the IoU matrix is fixture input)."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "matching_hungarian_reference.npz"))
TAGS = ["h0", "h1", "h2", "h3", "h4", "h5", "h6"]


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("thr", [0.3, 0.5])
def test_hungarian_assignment_matches_reference(tag, thr):
    from liso_amd.kabsch.box_groundtruth_matching_iou import hungarian_match_iou_matrix

    ig, ip, d, pm, gm = hungarian_match_iou_matrix(G[f"{tag}_iou"], thr)
    assert np.array_equal(ig, G[f"{tag}_{thr}_idx_gt"]) and np.array_equal(ip, G[f"{tag}_{thr}_idx_pred"])
    assert np.allclose(d, G[f"{tag}_{thr}_dists"], rtol=0, atol=0)
    assert np.array_equal(pm, G[f"{tag}_{thr}_pred_mask"]) and np.array_equal(gm, G[f"{tag}_{thr}_gt_mask"])


@pytest.mark.gpu
def test_hungarian_branch_on_device_boxes_is_the_optimal_assignment_of_the_hip_iou_matrix():
    from liso_amd.kabsch.box_groundtruth_matching_iou import hungarian_match_iou_matrix, match_boxes_by_descending_confidence_iou
    from liso_amd.kabsch.shape_utils import Shape
    from liso_amd.utils.nms_iou import box_iou_matrix

    g = torch.Generator().manual_seed(0)

    def boxes(n):
        pos = torch.cat([torch.rand(n, 2, generator=g) * 30 - 15, torch.zeros(n, 1)], -1)
        dims = torch.stack([torch.rand(n, generator=g) * 3 + 2, torch.rand(n, generator=g) + 1.5, torch.full((n,), 1.5)], -1)
        return Shape(pos=pos.cuda(), dims=dims.cuda(), rot=((torch.rand(n, 1, generator=g) * 2 - 1) * 3.14159).cuda(),
                     probs=torch.rand(n, 1, generator=g).cuda(), valid=torch.ones(n, dtype=torch.bool).cuda())

    gt, pred = boxes(40), boxes(70)
    got = match_boxes_by_descending_confidence_iou(gt, pred, 0.1, matching_mode="hungarian")
    want = hungarian_match_iou_matrix(box_iou_matrix(gt, pred, "iou_bev").cpu().numpy(), 0.1)
    assert all(np.array_equal(a, b) for a, b in zip(got, want)) and len(got[0]) > 0
    assert len(set(got[0].tolist())) == len(got[0]) and len(set(got[1].tolist())) == len(got[1])  # one-to-one
