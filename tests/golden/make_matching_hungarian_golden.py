"""Generates tests/golden/matching_hungarian_reference.npz from the reference's own
liso.kabsch.box_groundtruth_matching_iou.match_boxes_by_descending_confidence_iou(..., matching_mode="hungarian") (:70-118).
As in make_tracking_golden.py the IoU matrix is an INPUT of the fixture (the reference obtains it from its CUDA extension, which
cannot run here) and is handed to the function in place of box_iou_matrix.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_matching_hungarian_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import _Anything, import_with_stubs  # noqa: E402

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")


def main():
    def _imp():
        from liso.kabsch.shape_utils import Shape
        import liso.kabsch.box_groundtruth_matching_iou as matching
        return Shape, matching

    Shape, matching = import_with_stubs(_imp)
    g = np.random.default_rng(7)
    out = {}
    for tag, (n_gt, n_pred) in {"h0": (9, 14), "h1": (40, 25), "h2": (1, 6), "h3": (5, 1), "h4": (60, 60), "h5": (0, 4), "h6": (3, 0)}.items():
        iou = g.uniform(0.0, 1.0, (n_gt, n_pred)).astype(np.float32)
        iou[g.uniform(size=iou.shape) < 0.5] = 0.0
        if n_gt > 4 and n_pred > 4:
            iou[3, 2] = np.nan
            iou[4, 1] = np.inf
        matching.box_iou_matrix = lambda a, b, mode, _m=iou: torch.from_numpy(_m.copy())
        dummy = lambda n: Shape(pos=torch.zeros(n, 3), dims=torch.ones(n, 3), rot=torch.zeros(n, 1),  # noqa: E731
                                probs=torch.ones(n, 1), valid=torch.ones(n, dtype=torch.bool))
        for thr in (0.3, 0.5):
            ig, ip, d, pm, gm = matching.match_boxes_by_descending_confidence_iou(dummy(n_gt), dummy(n_pred), thr, matching_mode="hungarian")
            out.update({f"{tag}_{thr}_idx_gt": np.asarray(ig, dtype=np.int64), f"{tag}_{thr}_idx_pred": np.asarray(ip, dtype=np.int64),
                        f"{tag}_{thr}_dists": np.asarray(d, dtype=np.float64), f"{tag}_{thr}_pred_mask": pm, f"{tag}_{thr}_gt_mask": gm})
        out[f"{tag}_iou"] = iou
    dst = os.path.join(HERE, "matching_hungarian_reference.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, len(out), "arrays")


if __name__ == "__main__":
    main()
