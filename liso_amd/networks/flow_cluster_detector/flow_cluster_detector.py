"""Flow -> pseudo-boxes.  Mirror of liso/networks/flow_cluster_detector/flow_cluster_detector.py: `FlowClusterDetector`
(:14-336) and `fit_bev_box_z_and_height_using_points_in_box` (:339-384), built from the device stages BEV dynamicness
(D1), clustering + region moments (D2), z-fit (D3), Kabsch heading/velocity (D4-D6).  TensorBoard image logging
(:250-309) is not part of the path."""
import numpy as np
import torch

from liso_amd import _lib as L
from liso_amd.kabsch.shape_utils import Shape, extract_motion_in_pred_box_coordinates
from liso_amd.utils.bev_flow_utils import get_bev_dynamic_flow_map_from_pcl_flow_and_odom


@torch.no_grad()
def fit_bev_box_z_and_height_using_points_in_box(pcl, boxes: Shape, box_height=1000.0):
    """reference :339-384 -> (num_pts_in_box int64[K], fitted_box_z[K], fitted_box_height[K]) for one sample.
    One fused pass (lanes = boxes) instead of an [N,K,4] fp64 einsum."""
    assert len(pcl.shape) == 2, pcl.shape
    assert len(boxes.pos.shape) == 2, boxes.pos.shape
    L.require_cuda(pcl)
    dev = pcl.device
    pts = pcl.float().contiguous()
    K = boxes.pos.shape[0]
    num = torch.zeros(K, dtype=torch.int64, device=dev)
    fz = torch.zeros(K, dtype=torch.float32, device=dev)
    fh = torch.zeros(K, dtype=torch.float32, device=dev)
    if K == 0:
        return num, fz, fh
    pos, dims, rot = boxes.pos.float().contiguous(), boxes.dims.float().contiguous(), boxes.rot[..., 0].float().contiguous()
    lib = L.lib()
    nbytes = lib.liso_fit_box_z_workspace_bytes(pts.shape[0], K)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("fit_box_z", lambda: lib.liso_fit_box_z_f32(
            L.ptr(pts), pts.shape[-1], pts.shape[0], L.ptr(pos), pos.shape[-1], L.ptr(dims), dims.shape[-1], L.ptr(rot), K,
            float(box_height), L.ptr(num), L.ptr(fz), L.ptr(fh), L.ptr(ws), nbytes, L.stream_ptr())), "fit_box_z")
    return num, fz, fh


# ---- clustering block (reference :139-189) ---------------------------------------------------------------------------
@torch.no_grad()
def cluster_dynamic_pillars(dynamic_mask, bev_nonrigid_flow, row_coords_m, col_coords_m, eps=1.0, min_samples=5,
                            flow_similarity_importance=2.0, pitch=None):
    """DBSCAN(eps, min_samples) of the dynamic pillars in (x, y, w fx, w fy, w fz) -- reference :151-172, on the device.
    dynamic_mask [B,gx,gy] bool, bev_nonrigid_flow [B,gx,gy,3] -> (labels int32 [B,gx,gy] with 0 = background/noise and
    k = sklearn label k-1, num_labels int64 [B])."""
    L.require_cuda(dynamic_mask, bev_nonrigid_flow)
    B, gx, gy = dynamic_mask.shape
    dev = dynamic_mask.device
    dyn = dynamic_mask.to(torch.uint8).contiguous()
    flow = bev_nonrigid_flow.float().contiguous()
    assert flow.shape == (B, gx, gy, 3), flow.shape
    xs, ys = row_coords_m.float().contiguous(), col_coords_m.float().contiguous()
    if pitch is None:  # (two device->host reads; callers that know the grid pass it)
        pitch = min(float(xs[1] - xs[0]) if gx > 1 else eps, float(ys[1] - ys[0]) if gy > 1 else eps)
    cfg = L.DbscanCfg(B, gx, gy, int(eps / pitch) + 1, int(min_samples), float(eps), float(flow_similarity_importance))
    core = torch.empty((B, gx, gy), dtype=torch.uint8, device=dev)
    parent = torch.empty((B, gx, gy), dtype=torch.int32, device=dev)
    is_root = torch.empty((B, gx, gy), dtype=torch.int32, device=dev)
    labels = torch.empty((B, gx, gy), dtype=torch.int32, device=dev)
    lib = L.lib()
    import ctypes
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("dbscan_components", lambda: lib.liso_dbscan_components(
            ctypes.byref(cfg), L.ptr(dyn), L.ptr(xs), L.ptr(ys), L.ptr(flow), L.ptr(core), L.ptr(parent), L.ptr(is_root),
            L.stream_ptr())), "dbscan_components")
        from liso_amd.networks.flow_cluster_detector.mining_ops import inclusive_scan_i32
        rank = inclusive_scan_i32(is_root.view(B, -1)).view(B, gx, gy)  # (own scan: no memset node when captured into a hipGraph)
        L.check(L.TIMER.launch("dbscan_labels", lambda: lib.liso_dbscan_labels(
            ctypes.byref(cfg), L.ptr(dyn), L.ptr(xs), L.ptr(ys), L.ptr(flow), L.ptr(core), L.ptr(parent), L.ptr(rank),
            L.ptr(labels), L.stream_ptr())), "dbscan_labels")
    return labels, rank.view(B, -1)[:, -1].long()


@torch.no_grad()
def label_region_props(labels, max_labels):
    """skimage.measure.regionprops(label_img) for labels 1..max_labels (reference :175-189) ->
    float64 [B,max_labels,5] = (centroid_row, centroid_col, orientation, axis_major_length, axis_minor_length)"""
    L.require_cuda(labels)
    B, gx, gy = labels.shape
    lab = labels.to(torch.int32).contiguous()
    mom = torch.empty((B, max_labels, 6), dtype=torch.int64, device=lab.device)
    props = torch.empty((B, max_labels, 5), dtype=torch.float64, device=lab.device)
    lib = L.lib()
    nbytes = lib.liso_region_props_workspace_bytes(B, gx, gy, max_labels)  # (0: more labels than the LDS path holds -> atomics path)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=lab.device)
    with torch.cuda.device(lab.device):
        L.check(L.TIMER.launch("region_props", lambda: lib.liso_region_props_ws(
            L.ptr(lab), B, gx, gy, max_labels, L.ptr(mom), L.ptr(props), L.ptr(ws), nbytes, L.stream_ptr())), "region_props")
    return props


class FlowClusterDetector(torch.nn.Module):
    """reference :14-336 -- point flow -> dynamic BEV pillars -> DBSCAN clusters -> one box per cluster (moments),
    z/height from the points inside, plausibility filters, heading/velocity from a per-box Kabsch fit.
    Every stage runs on the device; the only host round trips are the two box counts (clusters, surviving boxes) that
    size the padded `Shape`."""

    def __init__(self, cfg, min_num_pts_per_box=10, max_box_len_m=7.0, aspect_ratio_max=4.0, min_box_area_m2=0.35,
                 min_box_volume_m3=0.5) -> None:
        super().__init__()
        from liso_amd.kabsch.kabsch_mask import KabschDecoder
        from liso_amd.utils.bev_utils import get_bev_setup_params

        tcfg = cfg.data.tracking_cfg
        if getattr(tcfg, "flow_cluster_detector_ignore_min_box_size_limits", False):  # reference :24-35
            min_box_area_m2, min_box_volume_m3 = 0.0, 0.0
        if getattr(tcfg, "flow_cluster_detector_ignore_max_box_size_limits", False):  # reference :36-46
            aspect_ratio_max, max_box_len_m = 1000.0, 1000.0
        self.min_box_area_m2, self.min_box_volume_m3 = min_box_area_m2, min_box_volume_m3
        self.min_num_pts_per_box, self.aspect_ratio_max, self.max_box_len_m = min_num_pts_per_box, aspect_ratio_max, max_box_len_m
        self.cfg = cfg
        (self.bev_range_m_np, self.img_grid_size_np, self.bev_pixel_per_meter_res_np, self.pcl_bev_center_coords_homog_np,
         torch_params) = get_bev_setup_params(cfg)
        for name, param in torch_params.items():
            self.register_parameter(name, torch.nn.Parameter(param, requires_grad=False))
        self.min_residual_flow_thresh_mps = 1.0  # 0.1 m displacement in 100 ms (reference :70)
        self.bev_img_grid_size = np.array(self.cfg.data.img_grid_size)
        self.kabsch_decoder = KabschDecoder(cfg)
        c = self.pcl_bev_center_coords_homog_np  # [gx, gy, 4]: pillar pitch for the DBSCAN window, known on the host
        dx = float(np.float32(c[1, 0, 0]) - np.float32(c[0, 0, 0])) if c.shape[0] > 1 else 1.0
        dy = float(np.float32(c[0, 1, 1]) - np.float32(c[0, 0, 1])) if c.shape[1] > 1 else 1.0
        self._pitch = min(dx, dy)
        self.last_num_labels = None
        # metric pillar centres per row / column as dense vectors (the kernels' inputs; slices of the [gx,gy,4] Parameter are strided)
        self.register_buffer("_row_coords", torch.from_numpy(np.ascontiguousarray(c[:, 0, 0])).float(), persistent=False)
        self.register_buffer("_col_coords", torch.from_numpy(np.ascontiguousarray(c[0, :, 1])).float(), persistent=False)

    @torch.no_grad()
    def forward(self, sample_data_ta, writer=None, writer_prefix: str = "", global_step: int = None, is_batched=True,
                capacity: int = None, odom_minus_eye=None) -> Shape:
        """`capacity` (extension): fixed number of cluster / box slots.  The padded Shape then always has `capacity` columns and
        the forward issues NO device->host read (the reference-shaped call reads the cluster count and the surviving-box count to
        size its arrays); valid boxes, their order and values are the same as long as the cluster count (`self.last_num_labels`,
        a device tensor the caller checks later) does not exceed `capacity`.
        `odom_minus_eye` (extension): inv(odom_ta_tb) - I as fp64 [B,4,4], computed by the caller (a library LU inverse: callers that
        replay this method from a hipGraph evaluate it eagerly in front of the replay)."""
        pcl = sample_data_ta["pcl_ta"]["pcl"]
        pcl_w_ground = sample_data_ta["pcl_full_w_ground_ta"]
        pillar_coors = sample_data_ta["pcl_ta"]["pillar_coors"]
        point_flow = sample_data_ta[self.cfg.data.flow_source]["flow_ta_tb"]
        odom_ta_tb = sample_data_ta[self.cfg.data.odom_source]["odom_ta_tb"]
        thresh = sample_data_ta["src_trgt_time_delta_s"] * self.min_residual_flow_thresh_mps
        if is_batched:
            pcl_is_valid = sample_data_ta["pcl_ta"]["pcl_is_valid"]
        else:  # reference :105-112
            pcl_is_valid = torch.ones_like(pcl[:, 0], dtype=torch.bool)[None]
            pcl, pcl_w_ground, pillar_coors = pcl[None], pcl_w_ground[None], pillar_coors[None]
            point_flow, odom_ta_tb, thresh = point_flow[None], odom_ta_tb[None], thresh[None]
        dev = pcl.device
        bev_dynamicness, bev_nonrigid_flow = get_bev_dynamic_flow_map_from_pcl_flow_and_odom(
            pcl_is_valid=pcl_is_valid, pcl=pcl, pillar_coors=pillar_coors, point_flow=point_flow, odom_ta_tb=odom_ta_tb,
            target_shape=self.bev_img_grid_size, return_nonrigid_bev_flow=True, odom_minus_eye=odom_minus_eye)
        dynamic_mask = torch.squeeze(bev_dynamicness, dim=-1) > thresh.to(dev)[..., None, None]
        centers = self.pcl_bev_center_coords_homog  # [gx,gy,4] float32; x depends on the row only, y on the column only
        labels, num_labels = cluster_dynamic_pillars(dynamic_mask, bev_nonrigid_flow, self._row_coords, self._col_coords,
                                                     pitch=self._pitch if capacity else None)
        self.last_bev_labels, self.last_num_labels = labels, num_labels
        B = labels.shape[0]
        if capacity:
            k_max = int(capacity)
        else:
            k_max = int(num_labels.max()) if B > 0 else 0  # host round trip 1: sizes the padded box arrays
        if k_max == 0:
            boxes = Shape.from_list_of_shapes([Shape.createEmpty().to_tensor().to(dev) for _ in range(B)], numeric_padding_value=0.0)
            return boxes if is_batched else boxes[0]
        props = label_region_props(labels, k_max)  # [B,K,5] float64
        from liso_amd.kabsch.shape_utils import Shape as _Shape
        from liso_amd.networks.flow_cluster_detector import mining_ops as MO

        # reference :176-206: one box per region (centroid -> pillar centre, axis lengths -> metres), one launch
        ppm = self.bev_pixel_per_meter_res_np
        center, dims2, rot1, dims2_f32, rot1_f32 = MO.boxes_from_regions(props, self._row_coords, self._col_coords, ppm)
        assert dims2.shape[-1] == 2, "otherwise box fitting will use bad box size from clustering!"
        # z / height from the points inside every box (reference :339-384), all samples into one set of outputs
        zbuf = torch.zeros(B * k_max * 16, dtype=torch.uint8, device=dev)
        num_pts = zbuf[:B * k_max * 8].view(torch.int64).view(B, k_max)
        fit_z = zbuf[B * k_max * 8:B * k_max * 12].view(torch.float32).view(B, k_max)
        fit_h = zbuf[B * k_max * 12:].view(torch.float32).view(B, k_max)
        lib = L.lib()
        for b in range(B):
            pts = pcl_w_ground[b].float().contiguous()  # [M, >= 3]: the kernel reads x, y, z with the row stride
            nbytes = lib.liso_fit_box_z_workspace_bytes(pts.shape[0], k_max)
            ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                L.check(L.TIMER.launch("fit_box_z", lambda: lib.liso_fit_box_z_f32(
                    L.ptr(pts), pts.shape[-1], pts.shape[0], L.ptr(center[b]), 2, L.ptr(dims2_f32[b]), 2, L.ptr(rot1_f32[b]), k_max,
                    1000.0, L.ptr(num_pts[b]), L.ptr(fit_z[b]), L.ptr(fit_h[b]), L.ptr(ws), nbytes, L.stream_ptr())), "fit_box_z")
        # plausibility filters + dropping the rejected boxes of every sample + zero padding (reference :208-248, :311), one launch
        arr = MO.filter_compact(num_labels, center, dims2, rot1, num_pts, fit_z, fit_h, min_points=self.min_num_pts_per_box,
                                aspect_ratio_max=self.aspect_ratio_max, max_box_len_m=self.max_box_len_m,
                                min_box_area_m2=self.min_box_area_m2, min_box_volume_m3=self.min_box_volume_m3,
                                park_invalid=bool(capacity))
        s_max = k_max if capacity else int(arr["counts"].max())  # host round trip 2

        def cut(t):
            return t if s_max == t.shape[1] else t[:, :s_max].contiguous()

        keep = cut(arr["valid"]).view(torch.bool)
        boxes = _Shape(pos=cut(arr["pos"]), dims=cut(arr["dims"]), rot=cut(arr["rot"]), probs=cut(arr["probs"]), velo=cut(arr["velo"]),
                       valid=keep, class_id=cut(arr["class_id"]), difficulty=cut(arr["difficulty"]))
        if s_max > 0:
            # adapt the rotation of the box to the direction of the flow (reference :312-331).  The slots beyond the surviving boxes do
            # not exist in the reference-shaped call: with fixed slots their Kabsch copies are parked 1000 km away, where their soft
            # mask is exactly 0 in fp32 and the background weight prod_s (1 - w_s) does not see them
            kboxes = _Shape(pos=cut(arr["kabsch_pos"]), dims=cut(arr["kabsch_dims"]), rot=cut(arr["kabsch_rot"])[..., None],
                            probs=boxes.probs, valid=keep)
            # (fixed slots: the kernel skips the parked ones -- same outputs, a quarter of the mask evaluations at 18 boxes in 64 slots)
            trafos = self.kabsch_decoder.trafos_from_point_flow_packed(
                point_cloud_ta=pcl, valid_mask_ta=pcl_is_valid, pointwise_flow_ta_tb=point_flow, pred_boxes_ta=kboxes,
                slot_count=arr["counts"] if capacity else None)
            MO.box_motion(trafos, boxes.pos, boxes.rot, boxes.velo)  # heading += atan2(t_y, t_x), speed = |t|, one launch
        if not is_batched:
            boxes = boxes[0]
            assert len(boxes.shape) == 1, boxes.shape
        return boxes
