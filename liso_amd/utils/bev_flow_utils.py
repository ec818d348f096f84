"""BEV dynamicness / non-rigid flow maps from point flow + odometry.  Mirror of liso/utils/bev_flow_utils.py:6-77
(same keyword-only signature and return values), computed by one fused gfx950 scatter pass
(include/liso_flow_cluster.h) instead of an fp64 einsum + two index-expanding scatter_add_ calls."""
import torch

from liso_amd import _lib as L


@torch.no_grad()
def get_bev_dynamic_flow_map_from_pcl_flow_and_odom(*, pcl_is_valid, pcl, pillar_coors, point_flow, odom_ta_tb,
                                                    target_shape, return_nonrigid_bev_flow=False, odom_minus_eye=None):
    L.require_cuda(pcl, point_flow)
    B, N = pcl_is_valid.shape
    h, w = int(target_shape[0]), int(target_shape[1])
    dev = pcl.device
    pts = pcl.float().contiguous()
    fl = point_flow.float().contiguous()
    assert fl.shape[-1] >= 3, fl.shape
    val = pcl_is_valid.to(torch.uint8).contiguous()
    coors = pillar_coors.to(torch.int32).contiguous()
    # bev_flow_utils.py:30-33: inv(odom) - I in fp64 [B,4,4]
    ome = odom_minus_eye if odom_minus_eye is not None else odometry_minus_identity(odom_ta_tb)
    dyn = torch.empty((B, h, w, 1), dtype=torch.float32, device=dev)
    nrf = torch.empty((B, h, w, 3), dtype=torch.float32, device=dev)
    lib = L.lib()
    nbytes = lib.liso_bev_dynamic_flow_workspace_bytes(B, h, w)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("bev_dynamic_flow", lambda: lib.liso_bev_dynamic_flow_f32(
            L.ptr(pts), pts.shape[-1], L.ptr(val), L.ptr(coors), L.ptr(fl), fl.shape[-1], L.ptr(ome), B, N, h, w,
            L.ptr(dyn), L.ptr(nrf), L.ptr(ws), nbytes, L.stream_ptr())), "bev_dynamic_flow")
    if return_nonrigid_bev_flow:
        return dyn, nrf
    return dyn


@torch.no_grad()
def odometry_minus_identity(odom_ta_tb):
    """inv(odom_ta_tb) - I in fp64 [B,4,4] (bev_flow_utils.py:30-33).  On the device: one launch of the cofactor-expansion kernel
    (liso_odom_inverse_minus_eye_f64) -- no library LU call, no singularity check that reads back to the host, capturable into a
    hipGraph; host tensors (CPU-side tests) take torch's LU inverse like the reference."""
    dev = odom_ta_tb.device
    if not odom_ta_tb.is_cuda:
        return (torch.linalg.inv_ex(odom_ta_tb.double()).inverse - torch.eye(4, device=dev, dtype=torch.float64)[None]).contiguous()
    m = odom_ta_tb.double().contiguous()
    out = torch.empty_like(m)
    with torch.cuda.device(dev):
        L.check(L.lib().liso_odom_inverse_minus_eye_f64(L.ptr(m), m.numel() // 16, L.ptr(out), L.stream_ptr()), "odom_inverse_minus_eye")
    return out
