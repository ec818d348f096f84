"""weighted_pc_alignment: weighted Kabsch between two clouds -> T[4,4] fp64.

Mirror of liso/slim/slim_loss/weighted_pc_alignment.py:10-80 (same signature and return values; the duplicate module
liso/weighted_pc_alignment/weighted_pc_alignment.py:54-141 is aliased in liso_amd/weighted_pc_alignment).  The moments
stay differentiable torch reductions (gradients flow to weights and clouds in the SLIM loss); the 3x3 solve is the
gfx950 symmetric_orthogonalization kernel.  The debugging try/except of the reference (:49-70) is not reproduced.
"""
import torch

from liso_amd.torch_symm_ortho import symmetric_orthogonalization

EPSILON = 1e-7


def weighted_pc_alignment(cloud_t0, cloud_t1, weights, use_epsilon_on_weights=False):
    assert cloud_t0.shape[1:] == (3,) and cloud_t1.shape[1:] == (3,), (cloud_t0.shape, cloud_t1.shape)
    assert len(weights.shape) == 1
    if use_epsilon_on_weights:  # reference :26-34
        weights = weights + EPSILON
        not_enough_points = (weights > 0).sum() < 3
    else:
        not_enough_points = (weights > 0).sum() < 3
        # `if not_enough_points: weights += EPSILON` without a host sync
        weights = weights + EPSILON * not_enough_points.to(weights.dtype)
    cum_wts = weights.sum(dim=-1)
    mx_wtd = (cloud_t0 * weights[..., None]).sum(dim=0) / cum_wts
    my_wtd = (cloud_t1 * weights[..., None]).sum(dim=0) / cum_wts
    Xc = cloud_t0 - mx_wtd[None, :]
    Yc = cloud_t1 - my_wtd[None, :]
    Sxy_wtd = (Yc * weights[..., None]).T @ Xc / cum_wts
    R = symmetric_orthogonalization(Sxy_wtd.to(torch.double))
    t = my_wtd.to(torch.double) - R @ mx_wtd.to(torch.double)
    R = torch.cat([R, torch.zeros((1, 3), dtype=R.dtype, device=R.device)], dim=0)
    t = torch.cat([t, torch.ones((1,), dtype=t.dtype, device=t.device)], dim=-1)
    T = torch.cat([R, t[:, None]], dim=-1)
    return T, not_enough_points
