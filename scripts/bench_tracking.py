"""Times the tracking / validation kernels (include/liso_tracking.h) with HIP events on the launch stream.
  python scripts/bench_tracking.py          (on the GPU box)"""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.kabsch.box_groundtruth_matching_iou import greedy_match_iou_matrix  # noqa: E402
from liso_amd.tracker.box_points import FP32_PRODUCT, FP64_PRODUCT, points_in_boxes  # noqa: E402


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    g = np.random.default_rng(0)
    out = []
    for N, K in ((120000, 100), (120000, 1000)):
        pos = np.concatenate([g.uniform(-45, 45, (K, 2)), g.uniform(-1.2, -0.6, (K, 1))], -1)
        dims = np.stack([g.uniform(3.0, 5.5, K), g.uniform(1.5, 2.4, K), g.uniform(1.4, 2.0, K)], -1)
        boxes7 = torch.from_numpy(np.concatenate([pos, dims, g.uniform(-3.1, 3.1, (K, 1))], -1)).float()[None].cuda()
        pts = torch.from_numpy(np.concatenate([g.uniform(-50, 50, (N, 2)), g.uniform(-2, 1, (N, 1))], -1)).float()[None].cuda()
        flow = torch.randn(1, N, 3, device="cuda")
        valid = torch.rand(1, N, device="cuda") > 0.1
        vu8 = valid.to(torch.uint8)
        for name, fn, nbytes in (
                ("count+mean_flow fp32 product", lambda: points_in_boxes(boxes7, pts, point_valid=vu8, flow=flow, precision=FP32_PRODUCT),
                 N * 25 + K * 28 + K * 16),
                ("count fp64 product", lambda: points_in_boxes(boxes7, pts, precision=FP64_PRODUCT), N * 12 + K * 28 + K * 4),
                ("mask fp64 product", lambda: points_in_boxes(boxes7, pts, want_mask=True, want_count=False, precision=FP64_PRODUCT),
                 N * 12 + K * 28 + N * K)):
            us = timed(fn)
            out.append({"kernel": "points_in_boxes", "case": name, "points": N, "boxes": K, "us": round(us, 1),
                        "algorithmic_MB": round(nbytes / 1e6, 2), "GB/s": round(nbytes / us / 1e3, 1)})
        # what the reference formulation costs on the same device (torch, [N,K] mask + [N,K,3] product)
        if K == 100:
            from liso_amd.kabsch.shape_utils import Shape
            sh = Shape(pos=boxes7[..., :3], dims=boxes7[..., 3:6], rot=boxes7[..., 6:], probs=torch.ones(1, K, 1, device="cuda"),
                       valid=torch.ones(1, K, dtype=torch.bool, device="cuda"))

            def torch_formulation():
                T = torch.linalg.inv(sh.get_poses()).float()
                homog = torch.cat([pts, torch.ones_like(pts[..., :1])], -1)
                pb = torch.einsum("bkij,bnj->bnki", T, homog)
                m = torch.all(pb[..., :3].abs() < 0.5 * sh.dims[:, None], dim=-1)
                return (flow[:, :, None, :] * valid[:, :, None, None].float() * m[..., None].float()).sum(1) / m.sum(1).clip(min=1.0)[..., None]
            out.append({"kernel": "torch formulation of tracking.py:2176-2185", "points": N, "boxes": K, "us": round(timed(torch_formulation, 10), 1)})
    for n_gt, n_pred in ((50, 300), (300, 1000)):
        iou = torch.rand(n_gt, n_pred, device="cuda")
        iou[torch.rand_like(iou) < 0.9] = 0
        order = torch.randperm(n_pred, device="cuda")
        out.append({"kernel": "match_greedy", "n_gt": n_gt, "n_pred": n_pred, "us": round(timed(lambda: greedy_match_iou_matrix(iou, order, 0.3)), 1)})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
