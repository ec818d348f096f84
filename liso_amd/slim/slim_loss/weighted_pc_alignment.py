"""weighted_pc_alignment: weighted Kabsch between two clouds -> T[4,4] fp64.

Mirror of liso/slim/slim_loss/weighted_pc_alignment.py:10-80 (same signature and return values; the duplicate module
liso/weighted_pc_alignment/weighted_pc_alignment.py:54-141 is aliased in liso_amd/weighted_pc_alignment).  The weighted
moments come from one fused, differentiable gfx950 reduction (fp64 sums; the reference reduces in fp32), the 3x3
solve is the gfx950 symmetric_orthogonalization kernel.  The debugging try/except of the reference (:49-70) is not reproduced.
"""
import torch

from liso_amd import _lib as L
from liso_amd.torch_symm_ortho import symmetric_orthogonalization


class _WeightedMoments(torch.autograd.Function):
    """16 fp64 sums (include/liso_kabsch.h) of two clouds and their weights, differentiable in all three.
    x, y [N,3] / w [N] -> [16], or batched x, y [B,N,3] / w [B,N] -> [B,16] (one launch for all fits)."""

    @staticmethod
    def forward(ctx, x, y, w):
        L.require_cuda(x, y, w)
        x, y, w = x.float().contiguous(), y.float().contiguous(), w.float().contiguous()
        batched = x.dim() == 3
        B, n = (x.shape[0], x.shape[1]) if batched else (1, x.shape[0])
        lib = L.lib()
        out = torch.empty((B, 16), dtype=torch.float64, device=x.device)
        if B > 0:
            nbytes = lib.liso_weighted_moments_workspace_bytes(B)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            with torch.cuda.device(x.device):
                L.check(lib.liso_weighted_moments_fwd_f32(L.ptr(x), L.ptr(y), L.ptr(w), B, n, L.ptr(out), L.ptr(ws), nbytes,
                                                          L.stream_ptr()), "weighted_moments_fwd")
        ctx.save_for_backward(x, y, w)
        ctx.dims = (B, n, batched)
        return out if batched else out[0]

    @staticmethod
    def backward(ctx, g):
        x, y, w = ctx.saved_tensors
        B, n, _ = ctx.dims
        need = ctx.needs_input_grad
        gx = torch.empty_like(x) if need[0] else None
        gy = torch.empty_like(y) if need[1] else None
        gw = torch.empty_like(w) if need[2] else None
        g = g.double().reshape(B, 16).contiguous()
        if B > 0:
            with torch.cuda.device(x.device):
                L.check(L.lib().liso_weighted_moments_bwd_f32(L.ptr(x), L.ptr(y), L.ptr(w), B, n, L.ptr(g),
                                                              L.ptr(gx) if gx is not None else None,
                                                              L.ptr(gy) if gy is not None else None,
                                                              L.ptr(gw) if gw is not None else None, L.stream_ptr()),
                        "weighted_moments_bwd")
        return gx, gy, gw


EPSILON = 1e-7


def weighted_pc_alignment(cloud_t0, cloud_t1, weights, use_epsilon_on_weights=False, valid_mask=None):
    """`valid_mask` (extension): rows to ignore stay in the arrays with weight exactly 0 (and finite coordinates)
    instead of being removed by the caller with boolean indexing -- same moments, no device->host sync."""
    assert cloud_t0.shape[1:] == (3,) and cloud_t1.shape[1:] == (3,), (cloud_t0.shape, cloud_t1.shape)
    assert len(weights.shape) == 1
    if valid_mask is not None:
        weights = torch.where(valid_mask, weights, 0.0)
    if use_epsilon_on_weights:  # reference :26-34
        weights = weights + EPSILON
        if valid_mask is not None:
            weights = torch.where(valid_mask, weights, 0.0)
        not_enough_points = (weights > 0).sum() < 3
    else:
        not_enough_points = (weights > 0).sum() < 3
        # `if not_enough_points: weights += EPSILON` without a host sync
        eps = EPSILON * not_enough_points.to(weights.dtype)
        weights = weights + (eps if valid_mask is None else eps * valid_mask.to(weights.dtype))
    # reference :36-47 (weighted means, centred clouds, (Yc * w)^T Xc / sum w) from one pass over the points
    mom = _WeightedMoments.apply(cloud_t0, cloud_t1, weights)
    cum_wts = mom[0]
    mx_wtd, my_wtd = mom[1:4] / cum_wts, mom[4:7] / cum_wts
    Sxy_wtd = (mom[7:16].view(3, 3) - cum_wts * my_wtd[:, None] * mx_wtd[None, :]) / cum_wts
    R = symmetric_orthogonalization(Sxy_wtd.to(torch.double))
    t = my_wtd.to(torch.double) - R @ mx_wtd.to(torch.double)
    R = torch.cat([R, torch.zeros((1, 3), dtype=R.dtype, device=R.device)], dim=0)
    t = torch.cat([t, torch.ones((1,), dtype=t.dtype, device=t.device)], dim=-1)
    T = torch.cat([R, t[:, None]], dim=-1)
    return T, not_enough_points


def batched_weighted_pc_alignment(cloud_t0, cloud_t1, weights, valid_mask, use_epsilon_on_weights=False):
    """`weighted_pc_alignment` for B independent fits of the same (padded) length in one set of launches:
    cloud_t0, cloud_t1 [B,N,3] (finite, padding rows arbitrary), weights [B,N], valid_mask [B,N] bool
    -> (T [B,4,4] fp64, not_enough_points [B] bool).  Same arithmetic per fit as the per-sample function."""
    assert cloud_t0.shape[2:] == (3,) and cloud_t1.shape == cloud_t0.shape and weights.shape == cloud_t0.shape[:2]
    weights = torch.where(valid_mask, weights, 0.0)
    if use_epsilon_on_weights:  # reference :26-34
        weights = torch.where(valid_mask, weights + EPSILON, 0.0)
        not_enough_points = (weights > 0).sum(dim=1) < 3
    else:
        not_enough_points = (weights > 0).sum(dim=1) < 3
        weights = weights + (EPSILON * not_enough_points.to(weights.dtype))[:, None] * valid_mask.to(weights.dtype)
    mom = _WeightedMoments.apply(cloud_t0, cloud_t1, weights)  # [B,16]
    cum = mom[:, 0:1]
    mx, my = mom[:, 1:4] / cum, mom[:, 4:7] / cum
    Sxy = (mom[:, 7:16].view(-1, 3, 3) - cum[:, :, None] * my[:, :, None] * mx[:, None, :]) / cum[:, :, None]
    R = symmetric_orthogonalization(Sxy)
    t = my - torch.einsum("bij,bj->bi", R, mx)
    B = R.shape[0]
    T = torch.cat([torch.cat([R, t[:, :, None]], dim=2),
                   torch.cat([torch.zeros((B, 1, 3), dtype=R.dtype, device=R.device),
                              torch.ones((B, 1, 1), dtype=R.dtype, device=R.device)], dim=2)], dim=1)
    return T, not_enough_points
