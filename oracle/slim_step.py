"""TEST INFRASTRUCTURE ONLY: the SLIM train step on the host CPU, for bench.py's cpu_baseline leg ("kind": "port").

The product's SLIM host modules are plain torch except for six HIP-backed ops.  `cpu_port()` temporarily swaps those
for CPU implementations that follow the reference's own formulation:
  pillar encoder      -> oracle/pillars.py              (mmdet3d voxel_generator + PillarFeatureNet + scatter)
  correlation lookup  -> explicit all-pairs volume + avg_pool2d + grid_sample   (liso/slim/model/raft_code/corr.py:6-46)
  symmetric orthogonalisation -> torch.linalg.svd (fp64)                         (liso/torch_symm_ortho/__init__.py:63)
  1-nearest neighbour -> scipy.spatial.cKDTree (exact; stands in for pynanoflann, knn_graph.py:57-70)
  BEV -> point gather -> torch advanced indexing                                 (slim_loss/static_aggregation.py:8-31)
  weighted moments    -> torch reductions                                         (slim_loss/weighted_pc_alignment.py:36-47)
This is a *timing* port of the step, not a parity oracle: SLIM parity is pinned by fixtures generated from the
reference (tests/golden/make_slim*_golden.py).
"""
import contextlib
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import kabsch as OK
from . import pillars as OP


class _CpuCorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        b, d, h, w = fmap1.shape
        corr = torch.matmul(fmap1.view(b, d, h * w).transpose(1, 2), fmap2.view(b, d, h * w)) / torch.sqrt(torch.tensor(d).float())
        corr = corr.reshape(b * h * w, 1, h, w)
        self.pyr = [corr]
        for _ in range(num_levels - 1):
            corr = F.avg_pool2d(corr, 2, stride=2)
            self.pyr.append(corr)

    def __call__(self, coords):
        r = self.radius
        coords = coords.permute(0, 2, 3, 1)
        b, h, w, _ = coords.shape
        out = []
        d = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
        for i, corr in enumerate(self.pyr):
            c = coords.reshape(b * h * w, 1, 1, 2) / 2 ** i + delta
            H, W = corr.shape[-2:]
            grid = torch.cat([2 * c[..., :1] / (W - 1) - 1, 2 * c[..., 1:] / (H - 1) - 1], dim=-1)
            out.append(F.grid_sample(corr, grid, align_corners=True).view(b, h, w, -1))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def _cpu_pillar_forward(self, pcl_t0, img_t0=None):
    lyr = self.pts_voxel_encoder.pfn_layers[0]
    bev, occ, _ = OP.pillar_forward([p.detach().cpu().numpy() for p in pcl_t0], lyr.linear.weight, lyr.norm.weight, lyr.norm.bias,
                                    lyr.norm.running_mean, lyr.norm.running_var, self.training,
                                    tuple(self.cfg.data.bev_range_m), tuple(self.cfg.data.img_grid_size),
                                    self.cfg.data.z_pillar_cutoff_value)
    return bev, occ


class _CpuKnnIndex:
    def __init__(self, ref, **kw):
        from scipy.spatial import cKDTree
        self.tree = cKDTree(ref.detach().cpu().numpy()[:, :3])

    def query(self, x, return_dist_sqr=False):
        d, i = self.tree.query(x.detach().cpu().numpy()[:, :3], k=1)
        idx = torch.from_numpy(i.astype(np.int64))
        return (idx, torch.from_numpy(d.astype(np.float32)) ** 2) if return_dist_sqr else idx


@torch.no_grad()
def _cpu_knn_graph(x, *, index=None, k, loop=False, **kw):
    idx = (index if isinstance(index, _CpuKnnIndex) else _CpuKnnIndex(index)).query(x)
    return idx[:, None]


class _CpuWeightedMoments:
    """the 16 weighted sums of weighted_pc_alignment.py:36-47 with differentiable torch ops"""

    @staticmethod
    def apply(x, y, w):
        xd, yd, wd = x.double(), y.double(), w.double()
        if xd.dim() == 2:
            return torch.cat([wd.sum()[None], (wd[:, None] * xd).sum(0), (wd[:, None] * yd).sum(0),
                              ((yd * wd[:, None]).T @ xd).reshape(-1)])
        return torch.cat([wd.sum(1, keepdim=True), (wd[..., None] * xd).sum(1), (wd[..., None] * yd).sum(1),
                          torch.einsum("bn,bni,bnj->bij", wd, yd, xd).reshape(xd.shape[0], 9)], dim=1)


def _cpu_grid_to_points(grid_data, pointwise_voxel_coordinates_fs, pointwise_valid_mask, default_value, plan=None):
    coors = torch.where(pointwise_valid_mask[..., None], pointwise_voxel_coordinates_fs,
                        torch.zeros_like(pointwise_voxel_coordinates_fs)).long()
    b = torch.arange(pointwise_valid_mask.shape[0])[:, None].expand(-1, pointwise_valid_mask.shape[1])
    data = grid_data[b, coors[..., 0], coors[..., 1]]
    return torch.where(pointwise_valid_mask[..., None], data, default_value)


@contextlib.contextmanager
def cpu_port():
    import liso_amd.networks.pcl_to_feature_grid.pcl_to_feature_grid as pp
    import liso_amd.slim.model.head_decoder as hd
    import liso_amd.slim.model.raft_mod as rm
    import liso_amd.slim.slim_loss.static_aggregation as sa
    import liso_amd.slim.slim_loss.knn_graph as kg
    import liso_amd.slim.slim_loss.knn_wrapper as kw
    import liso_amd.slim.slim_loss.weighted_pc_alignment as wpa

    saved = (pp.PointsPillarFeatureNetWrapper.forward, rm.CorrBlock, wpa.symmetric_orthogonalization, kw.knn_graph, kg.KnnIndex)
    saved_gather = (hd.batched_grid_data_to_pointwise_data, sa.batched_grid_data_to_pointwise_data)
    hd.batched_grid_data_to_pointwise_data = sa.batched_grid_data_to_pointwise_data = _cpu_grid_to_points
    saved_mom, wpa._WeightedMoments = wpa._WeightedMoments, _CpuWeightedMoments
    pp.PointsPillarFeatureNetWrapper.forward = _cpu_pillar_forward
    rm.CorrBlock = _CpuCorrBlock
    wpa.symmetric_orthogonalization = OK.symm_ortho
    kw.knn_graph = _cpu_knn_graph
    kg.KnnIndex = _CpuKnnIndex
    try:
        yield
    finally:
        (pp.PointsPillarFeatureNetWrapper.forward, rm.CorrBlock, wpa.symmetric_orthogonalization, kw.knn_graph, kg.KnnIndex) = saved
        hd.batched_grid_data_to_pointwise_data, sa.batched_grid_data_to_pointwise_data = saved_gather
        wpa._WeightedMoments = saved_mom


def timed_slim_step(cfg, state_dict, sample_t0, sample_t1):
    """one SLIM fwd+bwd(+RMSprop) step on the host cores; returns (seconds, loss)"""
    from liso_amd.trainer import SlimTrainer

    def to_cpu(s):
        return {k: (to_cpu(v) if isinstance(v, dict) else [t.cpu() for t in v] if isinstance(v, list) else v.cpu()) for k, v in s.items()}

    with cpu_port():
        tr = SlimTrainer(cfg, torch.device("cpu"))
        tr.net.load_state_dict({k: v.cpu() for k, v in state_dict.items()})
        s0, s1 = to_cpu(sample_t0), to_cpu(sample_t1)
        t0 = time.perf_counter()
        loss = tr.step(s0, s1)
        return time.perf_counter() - t0, float(loss)
