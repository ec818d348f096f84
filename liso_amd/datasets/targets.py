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
    center_bool_mask[B,H,W]) with rot = (sin, cos) (torch_dataset_commons.py:225-228).
    Device tensors go through the two-launch gfx950 kernel of include/liso_detector.h; host tensors (and the reference
    fixture test) through the torch formulation below, which materialises [B,K,H,W]."""
    if boxes_pos.is_cuda:
        return _render_center_targets_hip(boxes_pos, boxes_dims, boxes_rot, boxes_valid, grid_size, bev_range_m)
    return render_center_targets_torch(boxes_pos, boxes_dims, boxes_rot, boxes_valid, grid_size, bev_range_m)


def _render_center_targets_hip(boxes_pos, boxes_dims, boxes_rot, boxes_valid, grid_size, bev_range_m):
    import ctypes

    from liso_amd import _lib as L

    dev = boxes_pos.device
    B, K = boxes_valid.shape
    H, W = int(grid_size[0]), int(grid_size[1])
    cfg = L.TargetsCfg(B, K, H, W, float(bev_range_m[0]), float(bev_range_m[1]))
    pos, dims = boxes_pos.float().contiguous(), boxes_dims.float().contiguous()
    rot, val = boxes_rot[..., 0].float().contiguous(), boxes_valid.to(torch.uint8).contiguous()
    box_max = torch.empty((B, max(K, 1)), dtype=torch.float32, device=dev)
    out = {"probs": torch.empty((B, H, W, 1), dtype=torch.float32, device=dev),
           "dims": torch.empty((B, H, W, 3), dtype=torch.float32, device=dev),
           "pos": torch.empty((B, H, W, 3), dtype=torch.float32, device=dev),
           "rot": torch.empty((B, H, W, 2), dtype=torch.float32, device=dev)}
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("render_center_targets", lambda: L.lib().liso_render_center_targets_f32(
            ctypes.byref(cfg), L.ptr(pos), L.ptr(dims), L.ptr(rot), L.ptr(val), L.ptr(box_max), L.ptr(out["probs"]),
            L.ptr(out["dims"]), L.ptr(out["pos"]), L.ptr(out["rot"]), L.ptr(mask), L.stream_ptr())), "render_center_targets")
    out["center_bool_mask"] = mask.bool()
    return out


def render_center_targets_torch(boxes_pos, boxes_dims, boxes_rot, boxes_valid, grid_size, bev_range_m):
    """the torch formulation (checked against the reference fixture in tests/test_targets.py)"""
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
