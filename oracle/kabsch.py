"""TEST INFRASTRUCTURE ONLY: torch-CPU restatement of the Kabsch rows D4/D5 and of
liso.torch_symm_ortho / weighted_pc_alignment.  Pinned by tests/golden/kabsch_*.npz generated from the reference's
own python (tests/golden/make_kabsch_golden.py)."""
import math

import torch


def symm_ortho(a):
    """liso/torch_symm_ortho/__init__.py:63-64: R = U Vh of the fp64 SVD (differentiable through torch's own SVD
    backward, which equals the reference's analytic backward wherever the singular values are distinct from -each other)."""
    U, D, Vh = torch.linalg.svd(a.double())
    return U @ Vh


def weighted_pc_alignment(cloud_t0, cloud_t1, weights, use_epsilon_on_weights=False, eps=1e-7):
    """liso/slim/slim_loss/weighted_pc_alignment.py:10-80"""
    if use_epsilon_on_weights:
        weights = weights + eps
        nep = (weights > 0).sum() < 3
    else:
        nep = (weights > 0).sum() < 3
        if nep:
            weights = weights + eps
    cum = weights.sum(dim=-1)
    mx = (cloud_t0 * weights[..., None]).sum(dim=0) / cum
    my = (cloud_t1 * weights[..., None]).sum(dim=0) / cum
    Xc, Yc = cloud_t0 - mx[None], cloud_t1 - my[None]
    S = (Yc * weights[..., None]).T @ Xc / cum
    R = symm_ortho(S)
    t = my.double() - R @ mx.double()
    T = torch.eye(4, dtype=torch.float64)
    T = torch.cat([torch.cat([R, t[:, None]], dim=1), torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=torch.float64)], dim=0)
    return T, nep


def cauchy(x):
    """liso/kabsch/kabsch_mask.py:26-28"""
    return 0.5 + 1 / math.pi * torch.atan(x)


def soft_masks(pos, dims, rot, points, slope, scale, softness="cauchy"):
    """kabsch_mask.py:149-228: [B,S,N] soft inside-box weights (fp32 like the reference when shapes are fp32)."""
    f = cauchy if softness == "cauchy" else torch.sigmoid
    c, s = torch.cos(rot[..., 0]), torch.sin(rot[..., 0])
    d = points[:, None, :, :3] - pos[:, :, None, :]
    bx = c[..., None] * d[..., 0] + s[..., None] * d[..., 1]
    by = -s[..., None] * d[..., 0] + c[..., None] * d[..., 1]
    bz = d[..., 2]
    dd = dims * scale
    return (f(slope * (dd[..., 0, None] / 2 - bx.abs())) * f(slope * (dd[..., 1, None] / 2 - by.abs()))
            * f(slope * (dd[..., 2, None] / 2 - bz.abs())))


def kabsch_trafos(pos, dims, rot, points, valid, flow, slope=15.0, buffer=0.25, softness="cauchy"):
    """kabsch_mask.py:328-508 -> (T[B,S+1,4,4] f64, cum_wts[B,S+1], fg_w[B,S,N])"""
    pts = torch.where(valid[..., None], points[..., :3], torch.zeros(()))          # :20-22
    fl = torch.where(valid[..., None], flow[..., :2], torch.zeros(()))
    w_fg = soft_masks(pos, dims, rot, pts, slope, 1.0 - buffer, softness)
    w_out = soft_masks(pos, dims, rot, pts, slope, 1.0 + buffer, softness)
    bg = 1.0 - (1.0 - torch.prod(1.0 - w_out, dim=1, keepdim=True))                  # :370-372, mask_fusing.py:4-6
    w = torch.cat([w_fg, bg], dim=1) * valid[:, None, :].float()                     # :417-419
    x = pts.clone()
    x[..., 2] = 0.0                                                                  # :413
    y = x + torch.cat([fl, torch.zeros_like(fl[..., :1])], dim=-1)
    eps = 1e-12
    cum = w.sum(dim=-1)
    w = torch.where((cum < eps)[..., None], torch.tensor(eps), w)                    # :454-470
    cum = w.sum(dim=-1)
    mx = (x[:, None] * w[..., None]).sum(dim=2) / cum[..., None]
    my = (y[:, None] * w[..., None]).sum(dim=2) / cum[..., None]
    Xc, Yc = x[:, None] - mx[:, :, None], y[:, None] - my[:, :, None]
    S = torch.einsum("bsnc,bsnd->bscd", Yc * w[..., None], Xc) / cum[:, :, None, None]
    R = symm_ortho(S)
    t = my.double() - torch.einsum("bsoc,bsc->bso", R, mx.double())
    T = torch.zeros(S.shape[:2] + (4, 4), dtype=torch.float64)
    T[..., :3, :3], T[..., :3, 3], T[..., 3, 3] = R, t, 1.0
    # the returned fg weights are the ones computed BEFORE the padding rows are zeroed (:353-360 vs :417-419):
    # padding rows were mapped to the origin, so they carry the mask value of the point (0,0,0)
    return T, cum, w_fg
