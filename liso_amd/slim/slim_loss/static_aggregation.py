"""Static-flow aggregation.  Mirror of liso/slim/slim_loss/static_aggregation.py:8-110."""
import torch

from liso_amd.slim.slim_loss.weighted_pc_alignment import weighted_pc_alignment


def batched_grid_data_to_pointwise_data(grid_data, pointwise_voxel_coordinates_fs, pointwise_valid_mask, default_value):
    """reference :8-31 -- gather [B,H,W,C] at each point's pillar (invalid rows -> default), out of place"""
    assert len(grid_data.shape) == 4, grid_data.shape
    coors = torch.where(pointwise_valid_mask[..., None], pointwise_voxel_coordinates_fs,
                        torch.zeros_like(pointwise_voxel_coordinates_fs)).long()
    b = torch.arange(pointwise_valid_mask.shape[0], device=grid_data.device)[:, None].expand(-1, pointwise_valid_mask.shape[1])
    data = grid_data[b, coors[..., 0], coors[..., 1]]
    return torch.where(pointwise_valid_mask[..., None], data, torch.as_tensor(default_value, dtype=data.dtype, device=data.device))


def compute_batched_bev_static_aggregated_flow(pc, pointwise_voxel_coordinates_fs, pointwise_valid_mask, static_flow_bev,
                                               staticness_weights, voxel_center_metric_coordinates_bev,
                                               use_eps_for_weighted_pc_alignment: bool = False):
    """reference :34-110 -- per sample: Kabsch of (points, points + static flow) weighted by staticness, then the rigid
    flow field (T - I) applied to every BEV cell centre."""
    assert len(static_flow_bev.shape) == 4 and static_flow_bev.shape[-1] == 2
    flow3 = torch.cat([static_flow_bev, torch.zeros_like(static_flow_bev[..., :1])], dim=-1)
    pw_flow = batched_grid_data_to_pointwise_data(flow3, pointwise_voxel_coordinates_fs, pointwise_valid_mask, 0.0)
    pw_static = batched_grid_data_to_pointwise_data(staticness_weights[..., None], pointwise_voxel_coordinates_fs,
                                                    pointwise_valid_mask, 0.0)[..., 0]
    centers = voxel_center_metric_coordinates_bev
    grid_h = torch.cat([centers, torch.zeros_like(centers[..., :1]), torch.ones_like(centers[..., :1])], dim=-1)
    flows, Ts, neps = [], [], []
    for b in range(staticness_weights.shape[0]):
        m = pointwise_valid_mask[b]
        T, nep = weighted_pc_alignment(pc[b][m][..., :3], (pc[b][..., :3] + pw_flow[b])[m], pw_static[b][m],
                                       use_epsilon_on_weights=use_eps_for_weighted_pc_alignment)
        # (T - I) applied to every cell centre as fp64 broadcast multiply-adds; the reference's einsum (:88-99) is a
        # [4x4]x[4xHW] DGEMM that rocBLAS runs with a 128x128 tile: 28 ms per call at 512^2 (measured), 12 calls per step
        D = T - torch.eye(4, dtype=torch.float64, device=T.device)
        flows.append((D[:2, 0] * grid_h[..., 0:1] + D[:2, 1] * grid_h[..., 1:2] + D[:2, 2] * grid_h[..., 2:3]
                      + D[:2, 3] * grid_h[..., 3:4]).float())
        Ts.append(T)
        neps.append(nep)
    return torch.stack(flows, dim=0), torch.stack(Ts, dim=0), torch.stack(neps, dim=0)
