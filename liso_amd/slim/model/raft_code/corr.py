"""CorrBlock: 4-level correlation pyramid lookup.  Mirror of liso/slim/model/raft_code/corr.py:6-56 (same ctor and
__call__ signature, same output [B, L*(2r+1)^2, h, w] fp32 and channel order).

The reference builds the all-pairs volume (67 MB at 512^2 BEV, 1.07 GB at 1024^2) plus three pooled copies and calls
grid_sample four times per RAFT iteration.  Correlation, average pooling and bilinear sampling are linear in fmap2,
so the same numbers are obtained from the pooled *feature maps* (2.8 MB, L2 resident) by the on-the-fly gfx950
kernel of include/liso_slim.h; nothing of size (hw)^2 is ever stored, forward or backward.
"""
import ctypes

import torch
import torch.nn.functional as F

from liso_amd import _lib as L


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _CorrLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, radius, fmap1, *fmap2_levels):
        L.require_cuda(fmap1, coords)
        B, hw, D = fmap1.shape
        _, _, h, w = coords.shape
        cfg = L.CorrCfg(B, h, w, D, len(fmap2_levels), radius)
        W7 = 2 * radius + 1
        coords = coords.detach().float().contiguous()
        out = torch.empty((B, h, w, len(fmap2_levels) * W7 * W7), dtype=torch.float32, device=fmap1.device)
        with torch.cuda.device(fmap1.device):
            L.check(L.TIMER.launch("corr_lookup_fwd", lambda: L.lib().liso_corr_lookup_fwd_f32(
                ctypes.byref(cfg), L.ptr(fmap1), _ptr_array(fmap2_levels), L.ptr(coords), L.ptr(out), L.stream_ptr())),
                "corr_lookup_fwd")
        ctx.save_for_backward(coords, fmap1, *fmap2_levels)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, fmap1, *levels = ctx.saved_tensors
        cfg = ctx.cfg
        g = grad_out.float().contiguous()
        g1 = torch.empty((cfg.levels + 1,) + tuple(fmap1.shape), dtype=torch.float32, device=fmap1.device)
        g2 = [torch.zeros_like(l) for l in levels]
        with torch.cuda.device(fmap1.device):
            L.check(L.TIMER.launch("corr_lookup_bwd", lambda: L.lib().liso_corr_lookup_bwd_f32(
                ctypes.byref(cfg), L.ptr(fmap1), _ptr_array(levels), L.ptr(coords), L.ptr(g), L.ptr(g1), _ptr_array(g2),
                L.stream_ptr())), "corr_lookup_bwd")
        return (None, None, g1[0]) + tuple(g2)


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        assert radius <= 3, "the gfx950 lookup kernel holds the (2r+2)^2 integer patch in one wavefront (r <= 3)"
        self.num_levels, self.radius = num_levels, radius
        B, D, h, w = fmap1.shape
        # channels-last query features [B, hw, D]
        self.fmap1 = fmap1.float().permute(0, 2, 3, 1).reshape(B, h * w, D).contiguous()
        # avg_pool2d of the correlation volume over (h2, w2) (reference :20-21) == correlation with the pooled fmap2
        self.levels = []
        f2 = fmap2.float()
        for i in range(num_levels):
            if i > 0:
                f2 = F.avg_pool2d(f2, 2, stride=2)
            self.levels.append(f2.permute(0, 2, 3, 1).contiguous())

    def __call__(self, coords):
        out = _CorrLookup.apply(coords, self.radius, self.fmap1, *self.levels)  # [B,h,w,C] storage
        return out.permute(0, 3, 1, 2)  # logical [B,C,h,w] like the reference (:46), channels-last memory

    @staticmethod
    def corr(fmap1, fmap2):
        """reference :48-56 -- the explicit all-pairs volume (API parity / tests only; never used by __call__)"""
        batch, dim, ht, wd = fmap1.shape
        corr = torch.matmul(fmap1.view(batch, dim, ht * wd).transpose(1, 2), fmap2.view(batch, dim, ht * wd))
        return corr.view(batch, ht, wd, 1, ht, wd) / torch.sqrt(torch.tensor(dim).float())
