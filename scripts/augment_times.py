"""Box-snippet augmentation at BASELINE's size (120k-point sweep, 512^2 BEV, 100 m): device path (liso_amd/datasets/box_augmentation.py)
vs the CPU oracle (oracle/box_augment.py = the reference's numpy / scipy formulation), per sample.  Prints one JSON line.
    python scripts/augment_times.py [--samples 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50)
    args = ap.parse_args()
    from liso_amd import _lib as L
    from liso_amd.datasets.box_augmentation import BoxAugmenter, BoxSnippetDb, free_location_mask
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.kabsch.shape_utils import Shape
    from oracle import box_augment as ob
    from tests.test_gpu_box_augment import make_cfg

    dev = torch.device("cuda:0")
    G, R, N = 512, 100.0, 120000
    s0, _ = slim_pair(3, dev, n_points=N, grid=G, bev_range_m=R)
    pcl = s0["pcl_ta"]["pcl"][0]
    valid = s0["pcl_ta"]["pcl_is_valid"][0]
    pcl, coors = pcl[valid], s0["pcl_ta"]["pillar_coors"][0][valid]
    rs = np.random.default_rng(0)
    M = 3000
    counts = rs.integers(11, 600, M)
    pcls = [np.concatenate([rs.uniform(-1, 1, (c, 3)) * [2.2, 1.0, 0.8], rs.uniform(0, 1, (c, 1))], -1).astype(np.float32) for c in counts]
    boxes = Shape(pos=torch.zeros(M, 3), dims=torch.tensor([[4.4, 2.0, 1.6]]).repeat(M, 1), rot=torch.zeros(M, 1), probs=torch.ones(M, 1))
    box_cfg = {"max_num_objs": 15, "min_artificial_obj_velo": 1.0, "max_artificial_obj_velo": 3.0, "max_scale_delta": 0.2,
               "max_points_dropout": 0.25, "use_raydrop_augm": False}
    aug = BoxAugmenter(make_cfg(G, R, box_cfg), BoxSnippetDb({"pcl_in_box_cosy": pcls, "boxes": boxes}, dev), need_flow=True)
    flow = torch.zeros(pcl.shape[0], 3, device=dev)
    sample = {"pcl_ta": {"pcl": pcl, "pillar_coors": coors}, "pcl_full_w_ground_ta": pcl, "pcl_full_no_ground_ta": pcl, "gt": {},
              "slim_flow": {"flow_ta_tb": flow}}
    np.random.seed(0)
    torch.manual_seed(0)
    for _ in range(5):
        aug.create_augmented_sample_from_box_snippet_db(0.1, sample)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.samples):
        aug.create_augmented_sample_from_box_snippet_db(0.1, sample)
    torch.cuda.synchronize()
    gpu_ms = 1e3 * (time.perf_counter() - t0) / args.samples
    aug.reference_draws = False
    t0 = time.perf_counter()
    for _ in range(args.samples):
        aug.create_augmented_sample_from_box_snippet_db(0.1, sample)
    torch.cuda.synchronize()
    gpu_fast_ms = 1e3 * (time.perf_counter() - t0) / args.samples
    # the mask kernels alone (HIP events)
    L.TIMER.enable_all()
    L.TIMER.reset()
    for _ in range(20):
        free_location_mask(coors, (G, G), 10)
    torch.cuda.synchronize()
    mask_us = 1e3 * float(np.mean(L.TIMER.durations_ms("bev_free_mask")))
    L.TIMER.disable_all()
    # CPU oracle on the same sweep
    db = {"points": pcls, "dims": boxes.dims.numpy(), "pos_z": boxes.pos[:, 2].numpy()}
    p_np, c_np, f_np = pcl.cpu().numpy(), coors.cpu().numpy(), flow.cpu().numpy()
    centers = ob.bev_center_coords([R, R], [G, G])
    n_cpu = max(3, args.samples // 10)
    t0 = time.perf_counter()
    for _ in range(n_cpu):
        ob.augment(p_np, c_np, f_np, db, [R, R], [G, G], box_cfg, centers=centers)
    cpu_ms = 1e3 * (time.perf_counter() - t0) / n_cpu
    t0 = time.perf_counter()
    ob.free_location_mask(c_np, (G, G), 10)
    cpu_mask_ms = 1e3 * (time.perf_counter() - t0)
    print(json.dumps({"workload": "box-snippet augmentation, 120k-pt sweep, 512x512 BEV, <= 15 pasted objects (targets at 128x128 on the device path only)",
                      "device_ms_per_sample": round(gpu_ms, 3), "device_ms_per_sample_fast_location_draws": round(gpu_fast_ms, 3), "free_mask_kernels_us": round(mask_us, 1),
                      "free_mask_algorithmic_bytes": int(coors.shape[0] * 8 + G * G), "cpu_oracle_ms_per_sample": round(cpu_ms, 2),
                      "cpu_oracle_free_mask_ms": round(cpu_mask_ms, 2), "cpu_threads": 1}))


if __name__ == "__main__":
    main()
