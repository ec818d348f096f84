"""KabschDecoder: soft box masks + per-box weighted Kabsch from point flow.

Mirror of liso/kabsch/kabsch_mask.py:231-508 (class name, the 4 non-trainable Parameters from get_bev_setup_params,
forward(), get_kabsch_trafos_from_point_flow() with the same keyword arguments and 5 return values).  The
point-cloud path (batched_padded_points given) runs as ONE fused gfx950 pass (include/liso_kabsch.h) instead of
materialising [B,S,N,4] box coordinates and four [B,S,N,3] products.  It is a no-grad path, exactly like its only
hot-path caller (FlowClusterDetector.forward runs under no_grad).
"""
import ctypes
from typing import Tuple

import numpy as np
import torch

from liso_amd import _lib as L
from liso_amd.kabsch.shape_utils import Shape
from liso_amd.utils.bev_utils import get_bev_setup_params


def cauchy(logits):
    """reference :26-28"""
    return 0.5 + 1 / np.pi * torch.atan(logits)


def get_mask_softness_fun(softness_fun):
    return {"cauchy": cauchy, "sigmoid": torch.sigmoid}[softness_fun]


class KabschDecoder(torch.nn.Module):
    def __init__(self, cfg, img_grid_size=None) -> None:
        super().__init__()
        self.cfg = cfg
        (self.bev_range_m_np, self.img_grid_size_np, self.bev_pixel_per_meter_res_np,
         self.pcl_bev_center_coords_homog_np, torch_params) = get_bev_setup_params(cfg)
        for name, param in torch_params.items():  # reference :244-251: Parameters, hence in the state_dict
            self.register_parameter(name, torch.nn.Parameter(param, requires_grad=False))
        self.softness_name = cfg.mask_rendering.softness_fun
        self.softness_fun = get_mask_softness_fun(self.softness_name)

    def _run(self, shapes: Shape, points, valid, flow, slope, scale_fg, scale_bg, softness_name, want_weights, slot_count=None):
        L.require_cuda(points, shapes.pos)
        B, N = valid.shape
        S = shapes.pos.shape[1]
        dev = points.device
        pts = points.float().contiguous()
        fl = flow.float().contiguous()
        c = L.KabschCfg(B, N, S, pts.shape[-1], fl.shape[-1], float(slope), float(scale_fg), float(scale_bg),
                        {"cauchy": 0, "sigmoid": 1}[softness_name])
        lib = L.lib()
        pos = shapes.pos.float().contiguous()
        dims = shapes.dims.float().contiguous()
        rot = shapes.rot[..., 0].float().contiguous()
        val = valid.to(torch.uint8).contiguous()
        T = torch.empty((B, S + 1, 4, 4), dtype=torch.float64, device=dev)
        cum = torch.empty((B, S + 1), dtype=torch.float32, device=dev)
        w = torch.empty((B, S, N), dtype=torch.float32, device=dev) if want_weights else None
        nbytes = lib.liso_kabsch_workspace_bytes(ctypes.byref(c))
        ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            if slot_count is not None:
                assert slot_count.dtype == torch.int32 and slot_count.is_contiguous() and slot_count.numel() == B and slot_count.is_cuda
            L.check(L.TIMER.launch("kabsch_trafos", lambda: lib.liso_kabsch_trafos_counted_f32(
                ctypes.byref(c), L.ptr(pts), L.ptr(val), L.ptr(fl), L.ptr(pos), L.ptr(dims), L.ptr(rot),
                L.ptr(slot_count) if slot_count is not None else None, L.ptr(T),
                L.ptr(cum), L.ptr(w) if w is not None else None, L.ptr(ws), nbytes, L.stream_ptr())), "kabsch_trafos")
        return T, cum, w

    @torch.no_grad()
    def forward(self, shapes: Shape, batched_padded_points=None, batched_padded_is_valid_points=None, shape_name=None,
                sigmoid_slope=None, obj_dim_scale=1.0, softness_func=None):
        """reference :255-326 -- soft mask probabilities [B,S,N] of every point for every box slot."""
        slope = sigmoid_slope if sigmoid_slope is not None else self.cfg.mask_rendering.pred_sigmoid_slope
        name = self.softness_name if softness_func is None else ("sigmoid" if softness_func is torch.sigmoid else "cauchy")
        if batched_padded_points is None and batched_padded_is_valid_points is None:
            # reference :274-276: no points -> the masks are rendered on the BEV grid's pillar centres, [B,S,gx,gy].  The same kernel
            # with the grid cells as the "points" (every sample sees the same cells)
            centers = self.pcl_bev_center_coords_homog  # [gx, gy, 4]
            gx, gy = centers.shape[0], centers.shape[1]
            B = shapes.pos.shape[0]
            cells = centers[..., :3].reshape(1, gx * gy, 3).expand(B, -1, -1).contiguous()
            valid = torch.ones((B, gx * gy), dtype=torch.bool, device=cells.device)
            _, _, w = self._run(shapes, cells, valid, torch.zeros((B, gx * gy, 2), device=cells.device), slope, obj_dim_scale,
                                obj_dim_scale, name, True)
            return w.view(B, -1, gx, gy), None
        B, N = batched_padded_is_valid_points.shape
        zero_flow = torch.zeros((B, N, 2), device=batched_padded_points.device)
        _, _, w = self._run(shapes, batched_padded_points, batched_padded_is_valid_points, zero_flow, slope,
                            obj_dim_scale, obj_dim_scale, name, True)
        return w, None

    @torch.no_grad()
    def get_kabsch_trafos_from_point_flow(self, *, point_cloud_ta, valid_mask_ta, pointwise_flow_ta_tb,
                                          pred_boxes_ta: Shape, sigmoid_slope=None, obj_dim_scale_buffer=None,
                                          softness_func=None, return_weights=True) -> Tuple[torch.Tensor, ...]:
        """reference :328-399 -> (fg_T[B,S,4,4] f64, fg_w[B,S,N], fg_cum[B,S], bg_T[B,1,4,4] f64, bg_cum[B,1]).
        `return_weights=False` (extension): fg_w is None -- the [B,S,N] map (14 MB at 30 boxes x 120k points) is not written"""
        slope = sigmoid_slope if sigmoid_slope is not None else self.cfg.mask_rendering.pred_sigmoid_slope
        buf = obj_dim_scale_buffer if obj_dim_scale_buffer is not None else self.cfg.mask_rendering.obj_dim_scale_buffer
        name = self.softness_name if softness_func is None else ("sigmoid" if softness_func is torch.sigmoid else "cauchy")
        T, cum, w = self._run(pred_boxes_ta, point_cloud_ta, valid_mask_ta, pointwise_flow_ta_tb[:, :, 0:2], slope,
                              1.0 - buf, 1.0 + buf, name, bool(return_weights))
        S = pred_boxes_ta.pos.shape[1]
        return T[:, :S], w, cum[:, :S], T[:, S:], cum[:, S:]

    @torch.no_grad()
    def trafos_from_point_flow_packed(self, *, point_cloud_ta, valid_mask_ta, pointwise_flow_ta_tb, pred_boxes_ta: Shape,
                                      sigmoid_slope=None, obj_dim_scale_buffer=None, softness_func=None, slot_count=None):
        """get_kabsch_trafos_from_point_flow without the weight map, as ONE tensor fp64 [B, S+1, 4, 4]: slots 0..S-1 the per-box
        transforms, slot S the background transform (what liso_mine_box_motion of include/liso_box_mining.h takes)"""
        slope = sigmoid_slope if sigmoid_slope is not None else self.cfg.mask_rendering.pred_sigmoid_slope
        buf = obj_dim_scale_buffer if obj_dim_scale_buffer is not None else self.cfg.mask_rendering.obj_dim_scale_buffer
        name = self.softness_name if softness_func is None else ("sigmoid" if softness_func is torch.sigmoid else "cauchy")
        # (the kernel reads x, y, z / flow x, y with the row strides: no [..., :3] / [..., 0:2] copies)
        # `slot_count` (int32 [B], device): only the first slot_count[b] slots hold boxes, the rest are parked (liso_kabsch_trafos_counted_f32)
        T, _, _ = self._run(pred_boxes_ta, point_cloud_ta, valid_mask_ta, pointwise_flow_ta_tb, slope, 1.0 - buf, 1.0 + buf, name, False,
                            slot_count=slot_count)
        return T
