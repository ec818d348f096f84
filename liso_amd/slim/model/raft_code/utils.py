"""RAFT helpers, mirror of liso/slim/model/raft_code/utils.py (same names and semantics)."""
import torch
import torch.nn.functional as F


def upflow_n(flow, n=8, mode="bilinear"):
    """reference :5-7 -- x n bilinear upsampling (align_corners=True), flow scaled by n"""
    return n * F.interpolate(flow, size=(n * flow.shape[2], n * flow.shape[3]), mode=mode, align_corners=True)


def uplogits_n(logits, n=8, mode="bilinear"):
    """reference :10-12"""
    return F.interpolate(logits, size=(n * logits.shape[2], n * logits.shape[3]), mode=mode, align_corners=True)


def bilinear_sampler(img, coords, mode="bilinear", mask=False):
    """reference :15-28 -- grid_sample in pixel coordinates (kept for API parity; the hot path uses CorrBlock)"""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    grid = torch.cat([2 * xgrid / (W - 1) - 1, 2 * ygrid / (H - 1) - 1], dim=-1)
    img = F.grid_sample(img, grid, align_corners=True)
    if mask:
        m = (grid[..., :1] > -1) & (grid[..., 1:] > -1) & (grid[..., :1] < 1) & (grid[..., 1:] < 1)
        return img, m.float()
    return img


_GRIDS = {}


def coords_grid(batch, ht, wd, device):
    """reference :31-36 -- [B,2,ht,wd] with channel 0 = x (column), channel 1 = y (row).  A constant of (batch, ht, wd, device): built
    once (5 framework launches) and cloned per call (1) -- callers update their grid (coords1) or hold on to it (coords0)."""
    device = torch.device(device)
    key = (int(batch), int(ht), int(wd), device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else -1))
    g = _GRIDS.get(key)
    if g is None:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():  # (no constant born inside a capture is kept beyond it)
            ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
            return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)
        ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
        g = _GRIDS[key] = torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1).contiguous()
    return g.clone()


def initialize_flow(img, downscale_factor=8):
    """reference :40-46"""
    N, _, H, W = img.shape
    return coords_grid(N, H // downscale_factor, W // downscale_factor, device=img.device)
