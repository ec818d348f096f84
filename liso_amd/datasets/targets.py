"""CenterPoint training targets rendered on the device.

Follows liso/datasets/torch_dataset_commons.py:190-339 (draw_heat_regression_maps) and
liso/kabsch/kabsch_mask.py:56-116 (batched_render_gaussian_kabsch_mask): the reference renders these per sample in
DataLoader workers with numpy; on MI355X the detector step is tens of ms, so targets are rendered on the GPU from the
(pseudo-)boxes directly.
"""
import torch

from liso_amd.utils.bev_utils import get_metric_voxel_center_coords


def render_center_targets(boxes_pos, boxes_dims, boxes_rot, boxes_valid, grid_size, bev_range_m):
    """boxes_*: [B,K,3],[B,K,3],[B,K,1],[B,K] (padded) -> dict(probs[B,H,W,1], dims[..3], pos[..3], rot[..2],
    center_bool_mask[B,H,W]) with rot = (sin, cos) (torch_dataset_commons.py:225-228)."""
    dev = boxes_pos.device
    B, K = boxes_valid.shape
    H, W = int(grid_size[0]), int(grid_size[1])
    import numpy as np
    centers = torch.from_numpy(get_metric_voxel_center_coords(bev_range_m[0], bev_range_m[1], np.array([H, W]))[..., :2]
                               ).to(dev, torch.float32)                                  # [H,W,2]
    d = centers[None, None] - boxes_pos[:, :, None, None, :2]                             # [B,K,H,W,2]
    c, s = torch.cos(boxes_rot[..., 0]), torch.sin(boxes_rot[..., 0])
    # (x-mu)^T R diag(1/(0.15 len), 1/(0.15 w)) R^T (x-mu)   (kabsch_mask.py:93-102: cov = R diag(.15 l,.15 w) R^-1)
    u = d[..., 0] * c[:, :, None, None] + d[..., 1] * s[:, :, None, None]
    v = -d[..., 0] * s[:, :, None, None] + d[..., 1] * c[:, :, None, None]
    fac = u * u / (0.15 * boxes_dims[:, :, None, None, 0]) + v * v / (0.15 * boxes_dims[:, :, None, None, 1])
    heat = torch.exp(-fac / 2)
    heat = heat / torch.clamp(heat.amax(dim=(-1, -2), keepdim=True), min=1e-5)            # kabsch_mask.py:111-115
    heat = heat * boxes_valid[:, :, None, None].float()
    occ = (heat > 0.01).float()[..., None]                                                # commons.py:212-215
    probs = heat.amax(dim=1)[..., None]
    hottest = (heat.amax(dim=1, keepdim=True) == heat).float()[..., None] * occ * boxes_valid[:, :, None, None, None].float()
    sincos = torch.cat([torch.sin(boxes_rot), torch.cos(boxes_rot)], dim=-1)
    maps = {
        "probs": probs,
        "dims": (hottest * boxes_dims[:, :, None, None, :]).sum(1),
        "pos": (hottest * boxes_pos[:, :, None, None, :]).sum(1),
        "rot": (hottest * sincos[:, :, None, None, :]).sum(1),
    }
    # centre mask: the cell that contains the box centre (create_occupancy_pcl_image of the centres, commons.py:309-314)
    res = torch.tensor([H / bev_range_m[0], W / bev_range_m[1]], device=dev)
    ij = ((boxes_pos[..., :2] + torch.tensor(bev_range_m, device=dev) / 2) * res).long()
    ok = boxes_valid & (ij[..., 0] >= 0) & (ij[..., 0] < H) & (ij[..., 1] >= 0) & (ij[..., 1] < W)
    hits = torch.zeros((B, H * W), dtype=torch.int32, device=dev)
    lin = (ij[..., 0].clamp(0, H - 1) * W + ij[..., 1].clamp(0, W - 1))
    hits.scatter_add_(1, lin, ok.to(torch.int32))  # padded slots add 0: never clear a valid box's cell
    maps["center_bool_mask"] = (hits > 0).view(B, H, W)
    return maps
