"""Generates tests/golden/flow_cluster_reference.npz from the reference's own python:
  liso.utils.bev_flow_utils.get_bev_dynamic_flow_map_from_pcl_flow_and_odom
  liso.networks.flow_cluster_detector.flow_cluster_detector.fit_bev_box_z_and_height_using_points_in_box
Absent third-party packages are stubbed with empty modules (shapely, skimage, sklearn.cluster is present,
tensorboard): none of them is touched by the two functions above.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_flow_cluster_golden.py
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
for name in ("shapely", "shapely.affinity", "shapely.geometry", "skimage", "skimage.measure", "skimage.segmentation",
             "torch.utils.tensorboard", "torch.utils.tensorboard.writer", "tensorboard", "cv2", "matplotlib",
             "matplotlib.pyplot", "matplotlib.cm", "PIL", "PIL.Image", "torchvision", "torchvision.utils",
             "torchvision.transforms"):
    if name not in sys.modules:
        sys.modules[name] = MagicMock()


def main():
    from liso.kabsch.shape_utils import Shape
    from liso.networks.flow_cluster_detector.flow_cluster_detector import fit_bev_box_z_and_height_using_points_in_box
    from liso.utils.bev_flow_utils import get_bev_dynamic_flow_map_from_pcl_flow_and_odom

    g = torch.Generator().manual_seed(0)
    B, N, H, W, R = 2, 8000, 64, 64, 40.0
    pcl = torch.cat([torch.rand(B, N, 2, generator=g) * R - R / 2, torch.rand(B, N, 1, generator=g) * 3 - 2], -1)
    valid = torch.rand(B, N, generator=g) > 0.15
    pcl_nan = pcl.clone(); pcl_nan[~valid] = float("nan")
    coors = ((pcl[..., :2] + R / 2) / R * H).to(torch.int32)  # analyse_boxes.py:11-17 truncation convention
    coors[~valid] = -1
    flow = torch.randn(B, N, 3, generator=g) * 0.2
    flow[..., 2] = 0.0
    flow_nan = flow.clone(); flow_nan[~valid] = float("nan")
    th = torch.tensor([0.01, -0.02], dtype=torch.float64)
    odom = torch.eye(4, dtype=torch.float64).repeat(B, 1, 1)
    odom[:, 0, 0], odom[:, 0, 1], odom[:, 1, 0], odom[:, 1, 1] = torch.cos(th), -torch.sin(th), torch.sin(th), torch.cos(th)
    odom[:, 0, 3], odom[:, 1, 3] = torch.tensor([0.5, 0.9]), torch.tensor([0.05, -0.1])
    dyn, nrf = get_bev_dynamic_flow_map_from_pcl_flow_and_odom(
        pcl_is_valid=valid, pcl=pcl_nan, pillar_coors=coors, point_flow=flow_nan, odom_ta_tb=odom, target_shape=(H, W),
        return_nonrigid_bev_flow=True)
    out = dict(d1_pcl=pcl_nan.numpy(), d1_valid=valid.numpy(), d1_coors=coors.numpy(), d1_flow=flow_nan.numpy(),
               d1_odom=odom.numpy(), d1_dyn=dyn.numpy(), d1_nrf=nrf.numpy())
    # z-fit
    K = 9
    pts = torch.cat([torch.rand(6000, 2, generator=g) * 30 - 15, torch.rand(6000, 1, generator=g) * 3 - 2], -1)
    pos = torch.rand(K, 2, generator=g) * 24 - 12
    pos[-1] = torch.tensor([500.0, 500.0])  # a box without any point
    dims = torch.stack([torch.rand(K, generator=g) * 3 + 2, torch.rand(K, generator=g) + 1.2], -1)
    rot = (torch.rand(K, 1, generator=g) * 2 - 1) * 3.1
    boxes = Shape(pos=pos.clone(), dims=dims.clone(), rot=rot.clone(), probs=torch.ones(K, 1))
    num, fz, fh = fit_bev_box_z_and_height_using_points_in_box(pts, boxes, box_height=1000.0)
    out.update(d3_pts=pts.numpy(), d3_pos=pos.numpy(), d3_dims=dims.numpy(), d3_rot=rot.numpy(), d3_num=num.numpy(),
               d3_z=fz.numpy(), d3_h=fh.numpy())
    np.savez_compressed(os.path.join(HERE, "flow_cluster_reference.npz"), **out)
    print("dyn nonzero", int((dyn > 0).sum()), "num", num.tolist())


if __name__ == "__main__":
    main()
