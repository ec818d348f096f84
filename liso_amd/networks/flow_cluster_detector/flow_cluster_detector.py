"""Flow -> pseudo-boxes.  Mirror of liso/networks/flow_cluster_detector/flow_cluster_detector.py
(`fit_bev_box_z_and_height_using_points_in_box` now; the detector class is assembled in this module as its stages
land: BEV dynamicness (D1), clustering (D2), z-fit (D3), Kabsch heading/velocity (D4-D6))."""
import torch

from liso_amd import _lib as L
from liso_amd.kabsch.shape_utils import Shape


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
