"""CorrBlock: 4-level correlation pyramid lookup.  Mirror of liso/slim/model/raft_code/corr.py:6-56 (same ctor and
__call__ signature, same output [B, L*(2r+1)^2, h, w] fp32 and channel order).

The reference builds the all-pairs volume (67 MB at 512^2 BEV, 1.07 GB at 1024^2) plus three pooled copies and calls
grid_sample four times per RAFT iteration.  Correlation, average pooling and bilinear sampling are linear in fmap2,
so the same numbers are obtained from the pooled *feature maps* (2.8 MB, L2 resident) by the on-the-fly gfx950
kernel of include/liso_slim.h: the forward never stores anything of size (hw)^2.  The backward accumulates the window
gradients of all RAFT iterations into ONE dense volume-gradient per level (no float atomics, reproducible) and turns it
into feature gradients with two GEMMs per level, once per CorrBlock instead of once per iteration.
"""
import ctypes

import os

import torch
import torch.nn.functional as F

from liso_amd import _lib as L


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _VolumeGradState:
    """d loss / d (pooled correlation volumes) of one CorrBlock, accumulated over all lookups that used it"""

    def __init__(self):
        self.dvol = None


class _CorrFeatures(torch.autograd.Function):
    """Ties fmap1 / pooled fmap2 into the graph through a 1-element token.  Its backward runs exactly once, after the
    backward of every lookup of this CorrBlock has added its window gradients to `state.dvol` (autograd's dependency
    count on the token guarantees the order), and turns them into feature gradients with two GEMMs per level."""

    @staticmethod
    def forward(ctx, state, fmap1, *levels):
        ctx.state = state
        ctx.save_for_backward(fmap1, *levels)
        return fmap1.new_zeros(1)

    @staticmethod
    def backward(ctx, _grad_token):
        fmap1, *levels = ctx.saved_tensors
        dvol, ctx.state.dvol = ctx.state.dvol, None
        if dvol is None:
            return (None, torch.zeros_like(fmap1)) + tuple(torch.zeros_like(l) for l in levels)
        B, hw, D = fmap1.shape
        own = _own_gemms(fmap1, dvol, levels)
        if own is not None:
            return (None, own[0]) + tuple(own[1])
        g1 = None
        g2 = []
        for dv, f2 in zip(dvol, levels):
            f2m = f2.reshape(B, -1, D)
            g1 = torch.bmm(dv, f2m) if g1 is None else torch.baddbmm(g1, dv, f2m)
            g2.append(torch.bmm(dv.transpose(1, 2), fmap1).view_as(f2))
        return (None, g1) + tuple(g2)


def _own_gemms(fmap1, dvol, levels):
    """The two dense contractions of the correlation backward (liso/slim/model/raft_code/corr.py:48-56 is `fmap1^T fmap2`; its adjoint
    per level: d fmap1 += dvol . fmap2_l and d fmap2_l = dvol^T . fmap1) on the own MFMA kernels instead of rocBLAS batched GEMMs:
    per sample, `dvol . fmap2_l` is a 1x1 convolution over the hw query pixels with HW_l input channels and the pooled feature map as
    its [D, HW_l] filter, and `dvol^T . fmap1` is that convolution's WEIGHT gradient (sum over the query pixels of dvol[pixel, :]
    x fmap1[pixel, :]) -- fp32 tensors on the arithmetic of the process (F32X3 or exact fp32).  -> (g1 [B,hw,D], [g2_l like level l])
    or None when the kernels do not cover the shapes (channel counts must be multiples of 4)."""
    from liso_amd.utils import mfma_conv as MC

    B, hw, D = fmap1.shape
    if not fmap1.is_cuda or fmap1.dtype != torch.float32 or D % 4 or hw % 32:
        return None
    if os.environ.get("LISO_CORR_OWN_GEMM", "0") != "1":
        # measured on the SLIM step (120k points, 512^2): 15.6 ms with these launches vs 14.8 ms with the library's batched GEMMs --
        # a [4096 x 4096] . [4096 x 128] product as a 1x1 convolution is 64 blocks walking 128 channel slabs each; rocBLAS is 0.7 %
        # of the step.  Kept as a tested option (LISO_CORR_OWN_GEMM=1: no rocBLAS kernel in the step), off by default.
        return None
    rows = hw // 32  # (a 1x1 convolution does not care how the query pixels are arranged: rows of 32 = the kernels' tile width)
    if any(dv.shape[2] % 4 for dv in dvol):
        return None
    spec = MC.ConvSpec(1, 1, 1, 0, False)
    g1 = torch.empty_like(fmap1)
    g2 = [torch.empty_like(f2) for f2 in levels]
    for lvl, (dv, f2, out2) in enumerate(zip(dvol, levels, g2)):
        n = dv.shape[2]  # HW_l
        f2m = f2.reshape(B, n, D)
        for b in range(B):
            x = dv[b].view(1, rows, 32, n).permute(0, 3, 1, 2)                   # logical [1, C = HW_l, hw / 32, 32], channels last
            w = f2m[b].t().reshape(D, n, 1, 1).contiguous()                      # filter [D, HW_l, 1, 1]
            y, _ = MC.conv_forward(x, w, None, spec)                             # [1, D, hw / 32, 32], stored [1, hw / 32, 32, D]
            yb = y.permute(0, 2, 3, 1).reshape(hw, D)
            g1[b].copy_(yb) if lvl == 0 else g1[b].add_(yb)
            dyv = fmap1[b].view(1, rows, 32, D).permute(0, 3, 1, 2)              # "dy" [1, D, hw / 32, 32]
            res = MC.conv_wgrad(x, dyv, (D, n, 1, 1), spec, want_bias=False)     # dW[d, c] = sum_pixels dy[pixel, d] x[pixel, c]
            if res is None:
                return None
            out2.view(B, n, D)[b].copy_(res[0].view(D, n).t())
    return g1, g2


class _CorrLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, radius, state, token, fmap1, *fmap2_levels):
        L.require_cuda(fmap1, coords)
        B, hw, D = fmap1.shape
        _, _, h, w = coords.shape
        cfg = L.CorrCfg(B, h, w, D, len(fmap2_levels), radius)
        W7 = 2 * radius + 1
        coords = coords.detach().float().contiguous()
        out = torch.empty((B, h, w, len(fmap2_levels) * W7 * W7), dtype=torch.float32, device=fmap1.device)
        # the arithmetic of the process's fp32 convolutions: bf16 hi / lo pairs on the matrix cores ("x3": the tiled kernel, 4 x 8 queries
        # share the rows they correlate with) or fp32 FMAs ("exact": one wavefront per query and level)
        from liso_amd.utils import mfma_conv as MC

        tiled = MC.fp32_mode() == "x3" and os.environ.get("LISO_CORR_TILED", "1") != "0"
        fn = L.lib().liso_corr_lookup_fwd_tiled_f32 if tiled else L.lib().liso_corr_lookup_fwd_f32
        with torch.cuda.device(fmap1.device):
            L.check(L.TIMER.launch("corr_lookup_fwd_tiled" if tiled else "corr_lookup_fwd", lambda: fn(
                ctypes.byref(cfg), L.ptr(fmap1), _ptr_array(fmap2_levels), L.ptr(coords), L.ptr(out), L.stream_ptr())),
                "corr_lookup_fwd")
        ctx.save_for_backward(coords)
        ctx.cfg, ctx.state = cfg, state
        ctx.level_hw = [l.shape[1] * l.shape[2] for l in fmap2_levels]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (coords,) = ctx.saved_tensors
        cfg, st = ctx.cfg, ctx.state
        g = grad_out.float().contiguous()
        if st.dvol is None:
            st.dvol = [torch.zeros((cfg.batch, cfg.h * cfg.w, n), dtype=torch.float32, device=g.device) for n in ctx.level_hw]
        with torch.cuda.device(g.device):
            L.check(L.TIMER.launch("corr_lookup_bwd", lambda: L.lib().liso_corr_lookup_bwd_dvol_f32(
                ctypes.byref(cfg), L.ptr(coords), L.ptr(g), _ptr_array(st.dvol), L.stream_ptr())), "corr_lookup_bwd")
        return (None, None, None, g.new_zeros(1)) + (None,) * (1 + cfg.levels)


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
        self._state = _VolumeGradState()
        needs_grad = torch.is_grad_enabled() and (self.fmap1.requires_grad or any(l.requires_grad for l in self.levels))
        self._token = _CorrFeatures.apply(self._state, self.fmap1, *self.levels) if needs_grad else self.fmap1.new_zeros(1)
        self._fmap1_d = self.fmap1.detach()
        self._levels_d = [l.detach() for l in self.levels]

    def __call__(self, coords):
        out = _CorrLookup.apply(coords, self.radius, self._state, self._token, self._fmap1_d, *self._levels_d)  # [B,h,w,C]
        return out.permute(0, 3, 1, 2)  # logical [B,C,h,w] like the reference (:46), channels-last memory

    @staticmethod
    def corr(fmap1, fmap2):
        """reference :48-56 -- the explicit all-pairs volume (API parity / tests only; never used by __call__)"""
        batch, dim, ht, wd = fmap1.shape
        corr = torch.matmul(fmap1.view(batch, dim, ht * wd).transpose(1, 2), fmap2.view(batch, dim, ht * wd))
        return corr.view(batch, ht, wd, 1, ht, wd) / torch.sqrt(torch.tensor(dim).float())
