"""Host side of include/liso_box_mining.h: the per-box steps of the box mining as one-block kernels (no scan / sort library call,
no host read), shared by `FlowClusterDetector.forward` (flow_cluster_detector.py:176-248,312-331 of the reference) and
`perform_nms_on_shapes_padded` (nms_iou.py:23-66,257-282)."""
import ctypes

import torch

from liso_amd import _lib as L


def inclusive_scan_i32(x):
    """torch.cumsum(x, dim=1, dtype=int32) for int32 [B, n] on the device, without rocPRIM (its look-back state is memset: a memset
    node does not survive in a replayed hipGraph, liso_amd/utils/graph_safety.py)"""
    L.require_cuda(x)
    assert x.dtype == torch.int32 and x.dim() == 2 and x.is_contiguous()
    B, n = x.shape
    out = torch.empty_like(x)
    lib = L.lib()
    nbytes = lib.liso_scan_workspace_bytes(B, n)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        L.check(lib.liso_scan_inclusive_i32(L.ptr(x), B, n, L.ptr(out), L.ptr(ws), nbytes, L.stream_ptr()), "scan_inclusive_i32")
    return out


def boxes_from_regions(props, row_coords, col_coords, ppm):
    """props fp64 [B,K,5] -> (center fp32 [B,K,2], dims fp64 [B,K,2], rot fp64 [B,K], dims fp32, rot fp32)"""
    L.require_cuda(props)
    B, K, _ = props.shape
    dev = props.device
    props = props.contiguous()
    center = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
    dims = torch.empty((B, K, 2), dtype=torch.float64, device=dev)
    rot = torch.empty((B, K), dtype=torch.float64, device=dev)
    dims32 = torch.empty((B, K, 2), dtype=torch.float32, device=dev)
    rot32 = torch.empty((B, K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(L.lib().liso_mine_boxes_from_regions(
            L.ptr(props), B, K, L.ptr(row_coords), row_coords.shape[0], L.ptr(col_coords), col_coords.shape[0], float(ppm[0]),
            float(ppm[1]), L.ptr(center), L.ptr(dims), L.ptr(rot), L.ptr(dims32), L.ptr(rot32), L.stream_ptr()), "mine_boxes_from_regions")
    return center, dims, rot, dims32, rot32


def filter_compact(num_labels, center, dims2, rot, num_pts, fit_z, fit_h, *, min_points, aspect_ratio_max, max_box_len_m,
                   min_box_area_m2, min_box_volume_m3, park_invalid):
    """-> dict of the padded box arrays [B,K,...] (survivors first, label order kept), `counts` int32 [B] and the fp32 copies for
    the Kabsch kernel (kabsch_pos / kabsch_dims / kabsch_rot)"""
    B, K, _ = center.shape
    dev = center.device
    f32, f64, i32 = torch.float32, torch.float64, torch.int32
    o = {"pos": torch.empty((B, K, 3), dtype=f32, device=dev), "dims": torch.empty((B, K, 3), dtype=f64, device=dev),
         "rot": torch.empty((B, K, 1), dtype=f64, device=dev), "probs": torch.empty((B, K, 1), dtype=f64, device=dev),
         "velo": torch.empty((B, K, 1), dtype=f64, device=dev), "valid": torch.empty((B, K), dtype=torch.uint8, device=dev),
         "class_id": torch.empty((B, K, 1), dtype=i32, device=dev), "difficulty": torch.empty((B, K, 1), dtype=i32, device=dev),
         "counts": torch.empty((B,), dtype=i32, device=dev), "kabsch_pos": torch.empty((B, K, 3), dtype=f32, device=dev),
         "kabsch_dims": torch.empty((B, K, 3), dtype=f32, device=dev), "kabsch_rot": torch.empty((B, K), dtype=f32, device=dev)}
    cfg = L.MineFilterCfg(B, K, int(min_points), float(aspect_ratio_max), float(max_box_len_m), float(min_box_area_m2),
                          float(min_box_volume_m3), int(bool(park_invalid)))
    nl = num_labels.to(torch.int64).contiguous()
    with torch.cuda.device(dev):
        L.check(L.lib().liso_mine_filter_compact(
            ctypes.byref(cfg), L.ptr(nl), L.ptr(center), L.ptr(dims2), L.ptr(rot), L.ptr(num_pts), L.ptr(fit_z), L.ptr(fit_h),
            L.ptr(o["pos"]), L.ptr(o["dims"]), L.ptr(o["rot"]), L.ptr(o["probs"]), L.ptr(o["velo"]), L.ptr(o["valid"]),
            L.ptr(o["class_id"]), L.ptr(o["difficulty"]), L.ptr(o["counts"]), L.ptr(o["kabsch_pos"]), L.ptr(o["kabsch_dims"]),
            L.ptr(o["kabsch_rot"]), L.stream_ptr()), "mine_filter_compact")
    return o


def box_motion(trafos, pos, rot, velo):
    """trafos fp64 [B,S+1,4,4] (slot S = background); pos fp32 [B,S,3]; rot fp64 [B,S,1] updated in place; velo fp64 [B,S,1] written"""
    B, S1 = trafos.shape[:2]
    S = S1 - 1
    assert pos.shape == (B, S, 3) and pos.dtype == torch.float32 and pos.is_contiguous()
    assert rot.shape == (B, S, 1) and rot.dtype == torch.float64 and rot.is_contiguous() and velo.is_contiguous()
    assert trafos.dtype == torch.float64 and trafos.is_contiguous()
    with torch.cuda.device(pos.device):
        L.check(L.lib().liso_mine_box_motion(L.ptr(trafos), B, S, L.ptr(pos), L.ptr(rot), L.ptr(velo), L.stream_ptr()), "mine_box_motion")


def nms_select(arrays, max_num_boxes, overlap_threshold, pre_nms_max_num_boxes, targets_out=None):
    """arrays: dict(pos fp32 [B,K,3], dims fp64 [B,K,3], rot/probs/velo fp64 [B,K,1], valid uint8 [B,K], class_id / difficulty int32
    [B,K,1]), contiguous -- permuted and filtered IN PLACE.  -> (t_pos, t_dims, t_rot, t_valid): fp32 / uint8 arrays for the target
    renderer (`targets_out` = preallocated tuple to write into)."""
    from liso_amd import iou3d_nms_cuda

    pos = arrays["pos"]
    B, K, _ = pos.shape
    dev = pos.device
    lib = L.lib()
    dense = torch.empty((B, K, 7), dtype=torch.float32, device=dev)
    enters = torch.empty((B, K), dtype=torch.uint8, device=dev)
    nbytes = lib.liso_mine_nms_workspace_bytes(B, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a = arrays
    with torch.cuda.device(dev):
        L.check(lib.liso_mine_nms_prepare(
            B, K, int(pre_nms_max_num_boxes), L.ptr(a["pos"]), L.ptr(a["dims"]), L.ptr(a["rot"]), L.ptr(a["probs"]), L.ptr(a["velo"]),
            L.ptr(a["valid"]), L.ptr(a["class_id"]), L.ptr(a["difficulty"]), L.ptr(dense), L.ptr(enters), L.ptr(ws), nbytes,
            L.stream_ptr()), "mine_nms_prepare")
    if targets_out is None:
        targets_out = (torch.empty((B, K, 3), dtype=torch.float32, device=dev), torch.empty((B, K, 3), dtype=torch.float32, device=dev),
                       torch.empty((B, K), dtype=torch.float32, device=dev), torch.empty((B, K), dtype=torch.uint8, device=dev))
    t_pos, t_dims, t_rot, t_valid = targets_out
    for b in range(B):
        keep_dev, num_dev = iou3d_nms_cuda.nms_gpu_device(dense[b], overlap_threshold)
        with torch.cuda.device(dev):
            L.check(lib.liso_mine_nms_finish(
                b, K, int(max_num_boxes), L.ptr(keep_dev), L.ptr(num_dev), L.ptr(enters), L.ptr(a["pos"]), L.ptr(a["dims"]),
                L.ptr(a["rot"]), L.ptr(a["probs"]), L.ptr(a["velo"]), L.ptr(a["valid"]), L.ptr(a["class_id"]), L.ptr(a["difficulty"]),
                L.ptr(t_pos), L.ptr(t_dims), L.ptr(t_rot), L.ptr(t_valid), L.stream_ptr()), "mine_nms_finish")
    return t_pos, t_dims, t_rot, t_valid
