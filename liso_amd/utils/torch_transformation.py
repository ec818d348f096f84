"""Pose helpers, mirror of liso/utils/torch_transformation.py (same names and argument meaning).

torch_compose_matrix (reference :16-62) multiplies identity/translation/rotation 4x4s with einsum; the product is
written out directly here (T @ Rz), which yields the same fp64 values: every skipped factor is an exact 0 or 1.
"""
import numpy as np
import torch


def torch_decompose_matrix(matrix):
    """reference :5-13 -- (translation[...,3], theta_z[...,1]); refuses fp32 like the reference."""
    if matrix.dtype == torch.float32:
        raise UserWarning("You are decomposing a matrix with 32bit prec.", "This might  be unstable")
    translation = matrix[..., :3, 3].clone()
    theta_z = torch.atan2(matrix[..., 1, 0], matrix[..., 0, 0])
    return translation, theta_z[..., None]


def _compose(xp, t_x, t_y, theta_z, t_z, zeros, stack):
    if t_z is None:
        t_z = zeros(t_x)
    c, s = xp.cos(theta_z), xp.sin(theta_z)
    o, l = zeros(t_x), zeros(t_x) + 1.0
    rows = [stack([c, -s, o, t_x]), stack([s, c, o, t_y]), stack([o, o, l, t_z]), stack([o, o, o, l])]
    return rows


def torch_compose_matrix(t_x, t_y, theta_z, t_z=None):
    """reference :16-62 -- [B,S] scalars -> [B,S,4,4] (translation * rotation about z)."""
    assert t_x.shape == t_y.shape
    rows = _compose(torch, t_x, t_y, theta_z, t_z, torch.zeros_like, lambda v: torch.stack(v, dim=-1))
    return torch.stack(rows, dim=-2)


def numpy_compose_matrix(t_x, t_y, theta_z, t_z=None):
    """reference :65-110"""
    assert t_x.shape == t_y.shape
    rows = _compose(np, t_x, t_y, theta_z, t_z, np.zeros_like, lambda v: np.stack(v, axis=-1))
    return np.stack(rows, axis=-2)


def homogenize_pcl(pcl, is_not_padding=None):
    """reference :113-139 -- append w=1 (or validate an existing homogeneous coordinate)."""
    xp = torch if isinstance(pcl, torch.Tensor) else np
    if pcl.shape[-1] == 4:
        chk = pcl if is_not_padding is None else pcl[is_not_padding]
        assert bool(xp.all(xp.isfinite(chk))) if is_not_padding is not None else True
        assert bool(xp.all(chk[..., -1] == 1.0))
        return pcl
    ones = xp.ones_like(pcl[..., :1])
    return torch.cat([pcl, ones], dim=-1) if xp is torch else np.concatenate([pcl, ones], axis=-1)


def homogenize_flow(flow, is_not_padding=None):
    """reference :142-166 -- append w=0."""
    xp = torch if isinstance(flow, torch.Tensor) else np
    if flow.shape[-1] == 4:
        chk = flow if is_not_padding is None else flow[is_not_padding]
        assert bool(xp.all(chk[..., -1] == 0.0))
        return flow
    z = xp.zeros_like(flow[..., :1])
    return torch.cat([flow, z], dim=-1) if xp is torch else np.concatenate([flow, z], axis=-1)
