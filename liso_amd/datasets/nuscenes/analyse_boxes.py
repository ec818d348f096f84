"""Pillar coordinates of points, the dataset-side convention.  Mirror of liso/datasets/nuscenes/analyse_boxes.py:6-26
(`voxelize_pcl`; same name, arguments and results for numpy arrays and torch tensors on any device).

The convention is integer and therefore part of the bit-exact contract:
  coors = int32( (p + range/2) / range * grid )        -- conversion TRUNCATES toward zero, it is not a floor
so a point up to one pillar below the lower range limit lands in pillar 0 and counts as inside.  The arithmetic runs in
the promoted dtype of the operands exactly as numpy / torch promote them in the reference: float32 points with the
float64 range / int64 grid the caller `voxelize_sample` builds (torch_dataset_commons.py:975-987) give a float64
computation; an all-float32 call stays float32.
"""
import numpy as np
import torch


def voxelize_pcl(pcl_np, grid_range_m_np, grid_size):
    coors = (pcl_np[:, :3] + 0.5 * grid_range_m_np) / grid_range_m_np
    if torch.is_tensor(pcl_np):
        coors = (coors * grid_size).to(torch.int32)
    else:
        coors = (coors * grid_size).astype(np.int32)
    inside = ((0 <= coors[:, 0]) & (0 <= coors[:, 1]) & (0 <= coors[:, 2])
              & (coors[:, 0] < grid_size[0]) & (coors[:, 1] < grid_size[1]) & (coors[:, 2] < grid_size[2]))
    return coors, inside
