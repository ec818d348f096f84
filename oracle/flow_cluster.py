"""TEST INFRASTRUCTURE ONLY: torch-CPU restatement of rows D1 and D3.  Pinned by
tests/golden/flow_cluster_reference.npz generated from the reference's own python
(tests/golden/make_flow_cluster_golden.py)."""
import torch


def scatter_mean_2d(valid, batch_idx, rows, cols, values, B, H, W):
    """liso/utils/torch_differentiable_forward_scatter.py:22-87: scatter_add of values + counts, divide where count > 1"""
    C = values.shape[-1]
    v = values[valid].double()
    lin = ((batch_idx[valid] * H + rows[valid].long()) * W + cols[valid].long())
    tgt = torch.zeros(B * H * W, C, dtype=torch.float64)
    tgt.index_add_(0, lin, v)
    cnt = torch.zeros(B * H * W, dtype=torch.int64)
    cnt.index_add_(0, lin, torch.ones_like(lin))
    out = torch.where(cnt[:, None] > 1, tgt / cnt[:, None].clamp(min=1), tgt)
    return out.float().view(B, H, W, C)


def bev_dynamic_flow(pcl_is_valid, pcl, pillar_coors, point_flow, odom_ta_tb, target_shape):
    """liso/utils/bev_flow_utils.py:6-77 -> (dynamicness[B,H,W,1], nonrigid_flow[B,H,W,3])"""
    homog = torch.cat([pcl[..., :3], torch.ones_like(pcl[..., :1])], -1)
    homog = torch.where(pcl_is_valid[..., None], homog, torch.zeros(()))
    flow = torch.where(pcl_is_valid[..., None], point_flow, torch.zeros(()))
    M = torch.linalg.inv(odom_ta_tb.double()) - torch.eye(4, dtype=torch.float64)[None]
    stat = torch.einsum("bij,bnj->bni", M, homog.double())[..., :3].float()
    stat = torch.where(pcl_is_valid[..., None], stat, torch.zeros(()))
    nonrigid = flow - stat
    length = torch.linalg.norm(nonrigid, dim=-1, keepdim=True)
    B, N = pcl_is_valid.shape
    bidx = torch.arange(B)[:, None].repeat(1, N)
    H, W = int(target_shape[0]), int(target_shape[1])
    return (scatter_mean_2d(pcl_is_valid, bidx, pillar_coors[..., 0], pillar_coors[..., 1], length, B, H, W),
            scatter_mean_2d(pcl_is_valid, bidx, pillar_coors[..., 0], pillar_coors[..., 1], nonrigid, B, H, W))


def fit_box_z(pcl, pos, dims, rot, box_height=1000.0):
    """flow_cluster_detector.py:339-384"""
    K = pos.shape[0]
    c, s = torch.cos(rot.double()), torch.sin(rot.double())
    bz = pos[:, 2].double() if pos.shape[-1] == 3 else torch.zeros(K, dtype=torch.float64)
    dx = pcl[:, None, 0].double() - pos[None, :, 0].double()
    dy = pcl[:, None, 1].double() - pos[None, :, 1].double()
    lx, ly = (c * dx + s * dy).float(), (-s * dx + c * dy).float()
    lz = (pcl[:, None, 2].double() - bz[None]).float()
    d = dims if dims.shape[-1] == 3 else torch.cat([dims, box_height * torch.ones_like(dims[:, :1])], -1)
    inside = (lx.abs() < 0.5 * d[None, :, 0]) & (ly.abs() < 0.5 * d[None, :, 1]) & (lz.abs() < 0.5 * d[None, :, 2])
    zmax = torch.where(inside, lz, torch.tensor(-box_height)).max(dim=0).values
    zmin = torch.where(inside, lz, torch.tensor(box_height)).min(dim=0)
    h = torch.clip(zmax - zmin.values, min=1.0, max=2.0)
    return inside.sum(0), pcl[:, 2][zmin.indices] + 0.5 * h, h
