"""PointsPillarFeatureNetWrapper: list of [N_i, C] clouds -> (bev[B,64,H,W], occupancy[B,1,H,W]).

Mirror of liso/networks/pcl_to_feature_grid/pcl_to_feature_grid.py (same class name, ctor, forward signature and
state_dict keys `pts_voxel_encoder.pfn_layers.0.{linear.weight, norm.*}`).  The three mmcv/mmdet3d ops the
reference chains (Voxelization -> PillarFeatureNet -> PointPillarsScatter x2) are ONE fused gfx950 path here
(include/liso_pillars.h); the modules below only hold parameters under the reference's names.

The returned bev tensor has the reference's logical shape [B,64,H,W] (dim 2 = x index, dim 3 = y index,
pillar_scatter.py:87,97-99) but channels-last storage, which is what the BEV convolutions want on MI355X.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from liso_amd import _lib as L


class PFNLayer(nn.Module):
    """parameter holder for mmdet3d PFNLayer (voxel_encoders/utils.py:107-182): Linear(no bias) + BatchNorm1d."""

    def __init__(self, in_channels, out_channels, norm_cfg):
        super().__init__()
        self.units = out_channels
        self.norm = nn.BatchNorm1d(out_channels, eps=norm_cfg["eps"], momentum=norm_cfg["momentum"])
        self.linear = nn.Linear(in_channels, out_channels, bias=False)


class PillarFeatureNet(nn.Module):
    """parameter holder for mmdet3d PillarFeatureNet (voxel_encoders/pillar_encoder.py:13-91), single PFN layer,
    with_cluster_center=True, with_voxel_center=True, with_distance=False, legacy=True."""

    def __init__(self, in_channels, feat_channels, voxel_size, point_cloud_range, norm_cfg):
        super().__init__()
        assert len(feat_channels) == 1, "the LISO call site uses a single PFN layer (pcl_to_feature_grid.py:41-48)"
        self.in_channels = in_channels + 6
        self.pfn_layers = nn.ModuleList([PFNLayer(self.in_channels, feat_channels[0], norm_cfg)])
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range


def voxelize_raw(points, offsets, pcfg):
    """Deterministic hard voxelisation on the device (include/liso_pillars.h: liso_pillars_voxelize_f32).
    Returns fixed-stride products: coors[B*maxV,4], num_points[B*maxV], slots[B*maxV,20], num_voxels[B],
    cell_to_voxel[B*gx*gy] (row+1, 0 = empty cell)."""
    L.require_cuda(points)
    lib = L.lib()
    dev = points.device
    B = len(offsets) - 1
    n_total = int(offsets[-1])
    rows = B * pcfg.max_voxels
    i32 = dict(dtype=torch.int32, device=dev)
    coors = torch.empty((rows, 4), **i32)
    num_points = torch.empty((rows,), **i32)
    slots = torch.empty((rows, pcfg.max_points), **i32)
    num_voxels = torch.empty((B,), **i32)
    cell_to_voxel = torch.empty((B * pcfg.gx * pcfg.gy,), **i32)
    ws_bytes = lib.liso_pillars_voxelize_workspace_bytes(ctypes.byref(pcfg), B, n_total)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    off = (ctypes.c_int * (B + 1))(*[int(o) for o in offsets])
    with torch.cuda.device(dev):
        L.check(lib.liso_pillars_voxelize_f32(L.ptr(points), off, B, ctypes.byref(pcfg), L.ptr(coors),
                                              L.ptr(num_points), L.ptr(slots), L.ptr(num_voxels),
                                              L.ptr(cell_to_voxel), L.ptr(ws), ws_bytes, L.stream_ptr()),
                "pillars_voxelize")
    return coors, num_points, slots, num_voxels, cell_to_voxel


def decorate_rows(points, pcfg, B, coors, num_points, slots, num_voxels):
    """CSR feature rows of all kept points in pillar order (include/liso_pillars.h: liso_pfn_decorate_f32) ->
    (pt_off int32 [B*maxV+1], feat float32 [N,12], voxel_cell int32 [B*maxV])"""
    lib = L.lib()
    dev = points.device
    rows = B * pcfg.max_voxels
    pt_off = torch.empty(rows + 1, dtype=torch.int32, device=dev)
    feat = torch.empty((max(points.shape[0], 1), 12), dtype=torch.float32, device=dev)
    voxel_cell = torch.empty(rows, dtype=torch.int32, device=dev)
    nbytes = lib.liso_pfn_decorate_workspace_bytes(B, pcfg.max_voxels)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("pfn_decorate", lambda: lib.liso_pfn_decorate_f32(
            L.ptr(points), ctypes.byref(pcfg), B, L.ptr(coors), L.ptr(num_points), L.ptr(slots), L.ptr(num_voxels), L.ptr(pt_off),
            L.ptr(feat), L.ptr(voxel_cell), L.ptr(ws), nbytes, L.stream_ptr()), units=points.shape[0] * (points.shape[1] * 4 + 48)), "pfn_decorate")
    return pt_off, feat, voxel_cell


def pillar_prep(points, offsets, pcfg):
    """The weight-independent half of the pillar encoder: hard voxelisation + decorated feature rows (CSR) of a batch of clouds
    -> (pt_off, feat, voxel_cell, num_voxels, cell_to_voxel).  ~14 of the encoder's ~20 launches; a trainer may issue it ahead of the
    step that consumes it, on another stream (LisoLoopTrainer: the detector's clouds of the NEXT step, next to the current one)."""
    B = len(offsets) - 1
    coors, num_points, slots, num_voxels, cell_to_voxel = voxelize_raw(points, offsets, pcfg)
    pt_off, feat, voxel_cell = decorate_rows(points, pcfg, B, coors, num_points, slots, num_voxels)
    return pt_off, feat, voxel_cell, num_voxels, cell_to_voxel


class _PillarFeatureScatter(torch.autograd.Function):
    """voxelise (no grad) + fused PFN + scatter; differentiable w.r.t. linear.weight, norm.weight, norm.bias."""

    @staticmethod
    def forward(ctx, weight, gamma, beta, running_mean, running_var, points, offsets, pcfg, training, momentum, eps,
                out_dtype, out=None, prep=None):
        L.require_cuda(points, weight)
        lib = L.lib()
        dev = points.device
        B = len(offsets) - 1
        if prep is None:
            prep = pillar_prep(points, offsets, pcfg)
        pt_off, feat, voxel_cell, num_voxels, cell_to_voxel = prep
        weight = weight.contiguous().float()
        with torch.cuda.device(dev):
            st = L.stream_ptr()
            bn_out = torch.empty(4 * 64, dtype=torch.float32, device=dev)
            moments = torch.empty(80, dtype=torch.float64, device=dev)
            partials = torch.empty(lib.liso_pfn_partials_bytes(), dtype=torch.uint8, device=dev)
            L.check(lib.liso_pfn_bn_prepare_f32(L.ptr(feat), L.ptr(pt_off), ctypes.byref(pcfg), B, L.ptr(num_voxels),
                                                L.ptr(weight), L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                                                L.ptr(running_var), float(momentum), float(eps), int(training),
                                                L.ptr(bn_out), L.ptr(moments), L.ptr(partials), st), "pfn_bn_prepare")
            if out is not None:  # caller-owned destination (static graph inputs): [B, gx, gy, 64] rows + occupancy, both dense
                canvas, occupancy = out
                assert canvas.shape == (B, pcfg.gx, pcfg.gy, 64) and canvas.dtype == out_dtype and canvas.is_contiguous()
                assert occupancy.shape == (B, 1, pcfg.gx, pcfg.gy) and occupancy.is_contiguous()
            else:
                canvas = torch.empty((B, pcfg.gx, pcfg.gy, 64), dtype=out_dtype, device=dev)  # written densely by the kernel
                occupancy = torch.empty((B, 1, pcfg.gx, pcfg.gy), dtype=torch.float32, device=dev)
            L.check(L.TIMER.launch("pfn_forward_scatter", lambda: lib.liso_pfn_forward_scatter(
                L.ptr(feat), L.ptr(pt_off), L.ptr(voxel_cell), ctypes.byref(pcfg), B, L.ptr(cell_to_voxel), L.ptr(weight),
                L.ptr(bn_out), L.ptr(canvas), int(out_dtype == torch.bfloat16), L.ptr(occupancy), st),
                # algorithmic bytes: points read once + dense canvas (its own element size) + occupancy written once
                units=points.numel() * 4 + canvas.numel() * canvas.element_size() + occupancy.numel() * 4),
                "pfn_forward_scatter")
        ctx.save_for_backward(feat, pt_off, voxel_cell, num_voxels, weight, gamma, bn_out, moments)
        ctx.pcfg, ctx.B, ctx.training = pcfg, B, bool(training)
        ctx.mark_non_differentiable(occupancy)
        return canvas.permute(0, 3, 1, 2), occupancy

    @staticmethod
    def backward(ctx, grad_canvas, _grad_occ):
        feat, pt_off, voxel_cell, num_voxels, weight, gamma, bn_out, moments = ctx.saved_tensors
        lib = L.lib()
        dev = feat.device
        g = grad_canvas.permute(0, 2, 3, 1)
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        g = g.contiguous()
        F = weight.shape[1]
        gw = torch.empty((64, F), dtype=torch.float32, device=dev)
        gg = torch.empty(64, dtype=torch.float32, device=dev)
        gb = torch.empty(64, dtype=torch.float32, device=dev)
        partials = torch.empty(lib.liso_pfn_partials_bytes(), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.liso_pfn_backward(L.ptr(feat), L.ptr(pt_off), L.ptr(voxel_cell), ctypes.byref(ctx.pcfg), ctx.B,
                                          L.ptr(num_voxels), L.ptr(weight), L.ptr(gamma), L.ptr(bn_out), L.ptr(moments),
                                          int(ctx.training), L.ptr(g), int(g.dtype == torch.bfloat16), L.ptr(gw), L.ptr(gg),
                                          L.ptr(gb), L.ptr(partials), L.stream_ptr()), "pfn_backward")
        return gw, gg, gb, None, None, None, None, None, None, None, None, None, None, None


class PointsPillarFeatureNetWrapper(nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        self.cfg = cfg
        z_cut = cfg.data.setdefault("z_pillar_cutoff_value", 5.0)  # reference :14
        assert z_cut > 0.0, z_cut
        half = np.append(np.array(cfg.data.bev_range_m) / 2.0, z_cut)
        pc_range = np.concatenate([-half, half], axis=0)                      # reference :16-19
        voxel_size = np.append(np.array(cfg.data.bev_range_m) / np.array(cfg.data.img_grid_size), 2 * z_cut)
        self.max_num_points, self.max_voxels = 20, 40000                     # reference :24-30
        if "use_lidar_intensity" in cfg.data:
            num_input_channels = [3, 4][cfg.data.use_lidar_intensity]         # reference :31-35
        else:
            num_input_channels = 3
        # extension (north_star: (x, y, z, intensity, time) points of multi-sweep clouds): `data.num_point_channels: 5`
        # feeds the 5th channel to the PFN as one more decorated feature, exactly as mmdet3d's PillarFeatureNet treats
        # any extra input channel (pillar_encoder.py:93-159: features = [points, f_cluster, f_center])
        if cfg.data.setdefault("num_point_channels", None):
            num_input_channels = int(cfg.data.num_point_channels)
            assert num_input_channels in (3, 4, 5), num_input_channels
        self.num_input_channels = num_input_channels
        crf = cfg.network.centerpoint.setdefault("channel_reduction_factor", 1)
        assert 64 // crf == 64, "the fused gfx950 PFN kernel is built for 64 output channels"
        self.pts_voxel_encoder = PillarFeatureNet(
            in_channels=num_input_channels, feat_channels=[64 // crf], voxel_size=voxel_size,
            point_cloud_range=pc_range, norm_cfg={"type": "BN1d", "eps": 0.001, "momentum": 0.01})
        self.voxel_size, self.pc_range = voxel_size, pc_range
        self.grid = tuple(int(g) for g in cfg.data.img_grid_size)
        self.out_dtype = torch.float32  # set to torch.bfloat16 for the bf16 BEV backbone

    def _pcfg(self, n_channels):
        c = L.PillarCfg()
        c.x_min, c.y_min, c.z_min = (float(np.float32(v)) for v in self.pc_range[:3])
        c.vx, c.vy, c.vz = (float(np.float32(v)) for v in self.voxel_size)
        c.gx, c.gy = self.grid
        c.max_points, c.max_voxels, c.n_channels = self.max_num_points, self.max_voxels, n_channels
        return c

    @staticmethod
    def _cat(points):
        assert isinstance(points, (list, tuple)), type(points)
        offsets = [0]
        for p in points:
            assert p.dim() == 2, p.shape
            offsets.append(offsets[-1] + p.shape[0])
        cat = torch.cat([p.float() for p in points], dim=0).contiguous()
        return cat, offsets

    @torch.no_grad()
    def voxelize(self, points):
        """reference :56-84 -- (voxels[P,20,C], num_points[P], coors[P,4]=(b,0,x_idx,y_idx)), compacted.
        API-parity helper (the fused forward never builds `voxels`); compaction needs one host sync."""
        cat, offsets = self._cat(points)
        pcfg = self._pcfg(cat.shape[1])
        return self._compact(cat, pcfg, len(points), voxelize_raw(cat, offsets, pcfg))

    @staticmethod
    def _compact(cat, pcfg, B, vox):
        coors, num_points, slots, num_voxels, _ = vox
        nv = num_voxels.cpu().tolist()
        rows = torch.cat([torch.arange(b * pcfg.max_voxels, b * pcfg.max_voxels + nv[b], device=cat.device)
                          for b in range(B)])
        sl = slots[rows].long()
        valid = torch.arange(pcfg.max_points, device=cat.device)[None, :] < num_points[rows][:, None]
        vox = torch.where(valid[..., None], cat[sl.clamp(min=0, max=max(cat.shape[0] - 1, 0))],
                          torch.zeros((), device=cat.device))
        return vox, num_points[rows], coors[rows]

    @torch.no_grad()
    def prepare(self, pts):
        """(extension) voxelisation + decorated rows of `pts` now, for a later `extract_pts_feat(pts, prep=...)` (possibly on another
        stream: the caller orders the streams and keeps the result alive) -> opaque tuple"""
        cat, offsets = self._cat(pts)
        return (cat, offsets, pillar_prep(cat, offsets, self._pcfg(cat.shape[1])))

    def extract_pts_feat(self, pts, out=None, prep=None):
        """reference :86-102.  `out` (extension): (canvas rows [B, gx, gy, 64], occupancy [B, 1, gx, gy]) to write into;
        `prep` (extension): the result of `prepare(pts)` for the same clouds"""
        if prep is not None:
            cat, offsets, prep = prep
        else:
            cat, offsets = self._cat(pts)
        C = cat.shape[1]
        assert C == self.num_input_channels, (C, self.num_input_channels)
        lyr = self.pts_voxel_encoder.pfn_layers[0]
        training = self.training and lyr.norm.training
        if training and lyr.norm.track_running_stats:
            lyr.norm.num_batches_tracked += 1
        x, occ = _PillarFeatureScatter.apply(lyr.linear.weight, lyr.norm.weight, lyr.norm.bias, lyr.norm.running_mean,
                                             lyr.norm.running_var, cat, offsets, self._pcfg(C), training,
                                             lyr.norm.momentum, lyr.norm.eps, self.out_dtype, out, prep)
        return x, occ

    def forward(self, pcl_t0, img_t0=None, out=None, prep=None):
        """reference :104-107"""
        return self.extract_pts_feat(pcl_t0, out=out, prep=prep)
