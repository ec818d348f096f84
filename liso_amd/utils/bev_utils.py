"""BEV grid helpers, mirror of liso/utils/bev_utils.py:5-77."""
import numpy as np
import torch


def get_voxel_center_coords_m(bev_extent_m, net_output_shape_pix):
    """reference :24-40 -- metric centre of every BEV cell, [H,W,2] (ij indexing)."""
    shape = np.asarray(net_output_shape_pix)
    c = np.stack(np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing="ij"), axis=-1) + 0.5
    c = c / shape
    c = c * (bev_extent_m[2:] - bev_extent_m[:2])
    return c + bev_extent_m[:2]


def get_metric_voxel_center_coords(bev_range_x, bev_range_y, dataset_img_shape):
    """reference :5-21 -- homogeneous [H,W,4] = (x, y, 0, 1)."""
    ext = 0.5 * np.array([-bev_range_x, -bev_range_y, bev_range_x, bev_range_y])
    c = get_voxel_center_coords_m(bev_extent_m=ext, net_output_shape_pix=dataset_img_shape)
    return np.concatenate([c, np.zeros_like(c[..., :1]), np.ones_like(c[..., :1])], axis=-1)


def get_bev_setup_params(cfg):
    """reference :43-68"""
    bev_range_m_np = np.array(cfg.data.bev_range_m, np.float32)
    img_grid_size_np = np.array(cfg.data.img_grid_size).astype(np.int32)
    res = (img_grid_size_np / bev_range_m_np).astype(np.float32)
    centers = get_metric_voxel_center_coords(bev_range_x=bev_range_m_np[0], bev_range_y=bev_range_m_np[1],
                                             dataset_img_shape=img_grid_size_np).astype(np.float32)
    torch_params = {
        "bev_range_m": torch.from_numpy(bev_range_m_np),
        "bev_pixel_per_meter_resolution": torch.from_numpy(res),
        "img_grid_size": torch.from_numpy(img_grid_size_np),
        "pcl_bev_center_coords_homog": torch.from_numpy(centers),
    }
    return bev_range_m_np, img_grid_size_np, res, centers, torch_params
