"""Inner loops of box mining / tracking: mirrors of liso/tracker/tracking.py with the
points-in-boxes passes on the HIP kernel (include/liso_tracking.h)."""
import torch

from liso_amd.kabsch.shape_utils import Shape, extract_box_motion_transform_without_sensor_odometry
from liso_amd.tracker.box_points import FP32_PRODUCT, FP64_PRODUCT, dense_boxes, points_in_boxes


@torch.no_grad()
def mean_flow_per_box(pred_boxes: Shape, point_cloud_ta, valid_mask_ta, pointwise_flow_ta_tb):
    """reference tracking.py:2176-2185 -- [B,K,3] mean flow of the points inside each box, without the [B,N,K] mask and the
    [B,N,K,3] product.  As in the reference, `valid_mask_ta` removes a point's flow from the sum but not the point from the
    count."""
    res = points_in_boxes(dense_boxes(pred_boxes), point_cloud_ta[..., :3], point_valid=valid_mask_ta,
                          flow=pointwise_flow_ta_tb, precision=FP32_PRODUCT)
    return res["mean_flow"], res["count"]


@torch.no_grad()
def propagate_boxes_forward_using_flow(pred_boxes: Shape, point_cloud_ta, valid_mask_ta, pointwise_flow_ta_tb, odom_t0_t1, device,
                                       mean_flow=None):
    """reference tracking.py:2168-2211 (same arguments and return tuple).  `mean_flow` ([B,K,3], optional) reuses the
    per-box flow means of an earlier call: the tracker calls this twice per frame with +flow and -flow (:948-973), and the
    mean of -flow is minus the mean of flow."""
    if mean_flow is None:
        mean_flow, _ = mean_flow_per_box(pred_boxes, point_cloud_ta, valid_mask_ta, pointwise_flow_ta_tb)
    fg_kabsch_trafos = torch.eye(4, dtype=torch.float64, device=device)[None, None, ...].repeat(
        pred_boxes.shape[0], pred_boxes.shape[1], 1, 1)
    fg_kabsch_trafos[:, :, :3, 3] = mean_flow.double()
    bg_kabsch_trafo = torch.linalg.inv(odom_t0_t1)[None, None, ...].to(device)  # the odometry that fits the kabsch trafo
    bt0_deltaT_bt1 = extract_box_motion_transform_without_sensor_odometry(pred_boxes, fg_kabsch_trafos, bg_kabsch_trafo)
    st0_T_bt0 = pred_boxes.get_poses()
    st0_T_dyn_motion_warped_bt1 = (st0_T_bt0 @ bt0_deltaT_bt1)[0].detach().cpu()
    bg_kabsch_trafo = torch.linalg.inv(odom_t0_t1)[None, None, ...].to(device)
    st1_T_bt1 = fg_kabsch_trafos @ st0_T_bt0
    return fg_kabsch_trafos, odom_t0_t1, bg_kabsch_trafo, st0_T_dyn_motion_warped_bt1, st1_T_bt1


@torch.no_grad()
def count_points_in_boxes(pred_boxes: Shape, pcl_no_ground):
    """reference tracking.py:768-798 -- points of `pcl_no_ground` [N,>=3] inside each of the (unbatched) boxes [K]; BEV-only
    boxes get the reference's dummy height 2 m at z = -1 m."""
    boxes = pred_boxes.clone()
    if boxes.dims.shape[-1] == 2:
        boxes.dims = torch.cat([boxes.dims, 2.0 * torch.ones_like(boxes.dims[..., [0]])], dim=-1)
    if boxes.pos.shape[-1] == 2:
        boxes.pos = torch.cat([boxes.pos, -1.0 * torch.ones_like(boxes.dims[..., [0]])], dim=-1)
    res = points_in_boxes(dense_boxes(boxes)[None], pcl_no_ground[None, :, :3], precision=FP64_PRODUCT)
    return res["count"][0]


@torch.no_grad()
def drop_boxes_with_too_few_points(pred_boxes: Shape, pcl_no_ground, min_points_in_box: int):
    """reference tracking.py:768-812: `valid` <- at least `min_points_in_box` points inside, then drop the padding boxes"""
    num = count_points_in_boxes(pred_boxes, pcl_no_ground)
    out = pred_boxes.clone()
    out.valid = num >= min_points_in_box
    return out.drop_padding_boxes()


# ---- local box refinement (reference tracking.py:239-260, 2004-2133) -------------------------------------------------------------
@torch.no_grad()
def fit_boxes_to_points(pcl, boxes7, dims_bloat, point_valid=None):
    """Rectangle fit ("closeness to edge", liso/box_fitting/box_fitting.py:93-141,242-258) of the sweep points inside the bloated BEV
    footprint of every box: pcl [N,>=2] fp32 (cuda), boxes7 [K,7] -> (count int32 [K], fit float64 [K,5] = centre x, y, length,
    width, yaw; NaN rows where no point lies inside).  One launch for all boxes of the frame."""
    import ctypes  # noqa: F401

    from liso_amd import _lib as L

    L.require_cuda(pcl, boxes7)
    pts, b7 = pcl.float().contiguous(), boxes7.float().contiguous()
    n, k, dev = pts.shape[0], b7.shape[0], pts.device
    count = torch.zeros(k, dtype=torch.int32, device=dev)
    fit = torch.full((k, 5), float("nan"), dtype=torch.float64, device=dev)
    if k == 0 or n == 0:
        return count, fit
    nbytes = int(L.lib().liso_fit_boxes_closeness_workspace_bytes(n, k))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    valid = point_valid.to(torch.uint8).contiguous() if point_valid is not None else None
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("fit_boxes_closeness", lambda: L.lib().liso_fit_boxes_closeness_f32(
            L.ptr(pts), n, pts.shape[1], L.ptr(valid) if valid is not None else None, L.ptr(b7), k, float(dims_bloat), L.ptr(count),
            L.ptr(fit), L.ptr(ws), nbytes, L.stream_ptr())), "fit_boxes_closeness")
    return count, fit


def set_box_size_keep_closest_point_constant(boxes: Shape, new_box_dims) -> Shape:
    """reference tracking.py:239-260: give every box the size `new_box_dims`, keeping its bottom corner closest to the sensor in
    place (in-place on `boxes`, like the reference)"""
    boxes.assert_attr_shapes_compatible()
    corners, _ = boxes.get_box_corners()
    corners = corners[..., list(Shape.get_bottom_corner_idxs()), :]
    closest_idx = torch.argmin(torch.linalg.norm(corners[..., :2], dim=-1), dim=-1)
    closest = torch.gather(corners, -2, closest_idx[..., None, None].expand(*closest_idx.shape, 1, 3))[..., 0, :]
    shift_m = new_box_dims / boxes.dims * (boxes.pos - closest)
    boxes.pos = (closest + shift_m).to(boxes.pos.dtype) if closest.dtype != boxes.pos.dtype else closest + shift_m
    boxes.dims = torch.ones_like(boxes.dims) * new_box_dims
    boxes.assert_attr_shapes_compatible()
    return boxes


@torch.no_grad()
def perform_local_box_refinement(cfg, box_predictor, point_clouds_sensor_cosy, box_sequence_in_sensor_cosy_for_specific_track_id: Shape,
                                 track_age: int, start_time_idx: int):
    """reference tracking.py:2004-2133, same arguments and in-place semantics.  The per-frame body (points in the bloated footprint ->
    rectangle fit) is one kernel launch per frame and nothing is read back: frames without points inside keep their box through
    `torch.where` on the device, so a whole track is refined without a host synchronisation."""
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector

    seq = box_sequence_in_sensor_cosy_for_specific_track_id
    cfg.data.tracking_cfg.setdefault("box_refinement_dims_quantile", 0.95)
    box_dims_quantile = 0.95 if isinstance(box_predictor, FlowClusterDetector) else 0.6
    refined_box_dims = torch.quantile(seq.dims, q=box_dims_quantile, dim=0)
    fcfg = cfg.data.tracking_cfg.fit_box_to_points
    if fcfg.fit_rot or fcfg.fit_pos:
        assert seq.shape[0] == track_age, (seq.shape, track_age)
        dev = seq.pos.device
        rot = seq.rot if seq.rot is not None and seq.rot.shape[-1] > 0 else torch.zeros_like(seq.pos[..., :1])
        boxes7 = torch.cat([seq.pos, seq.dims, rot[..., :1]], dim=-1).float()
        for t in range(track_age):
            pcl = point_clouds_sensor_cosy[start_time_idx + t].to(dev)
            count, fit = fit_boxes_to_points(pcl, boxes7[t:t + 1], fcfg.fitting_dims_bloat_factor)
            has = count[0] > 0
            if fcfg.fit_rot:
                delta = (fit[0, 4] - seq.rot[t, 0].double()).to(seq.rot.dtype)  # refined yaw - float(box.rot), added as a scalar
                seq.rot[t, 0] = torch.where(has, seq.rot[t, 0] + delta, seq.rot[t, 0])
            if fcfg.fit_pos:
                seq.pos[t, :2] = torch.where(has, fit[0, :2].to(seq.pos.dtype), seq.pos[t, :2])
    return set_box_size_keep_closest_point_constant(seq, refined_box_dims)
