"""CorrBlock: 4-level correlation pyramid lookup.  Mirror of liso/slim/model/raft_code/corr.py:6-56 (same ctor and
__call__ signature, same output [B, L*(2r+1)^2, h, w] fp32 and channel order).

The reference builds the all-pairs volume (67 MB at 512^2 BEV, 1.07 GB at 1024^2) plus three pooled copies and calls
grid_sample four times per RAFT iteration.  Correlation, average pooling and bilinear sampling are linear in fmap2,
so the same numbers are obtained from the pooled *feature maps* (2.8 MB, L2 resident) by the on-the-fly gfx950
kernel of include/liso_slim.h: the forward never stores anything of size (hw)^2.  The backward accumulates the window
gradients of all RAFT iterations into ONE dense volume-gradient per level (no float atomics, reproducible) and turns it
into feature gradients with two launches of the library's own batched matrix-core kernel (all levels in each), once per CorrBlock
instead of once per iteration.  The pooled pyramid and its adjoint are one launch each.
"""
import ctypes
import os

import torch

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
        g1, g2 = corr_bwd_features(fmap1, levels, dvol)
        return (None, g1) + tuple(g2)


def corr_bwd_features(fmap1, levels, dvol):
    """The adjoint of the correlation (liso/slim/model/raft_code/corr.py:48-56 is `fmap1^T fmap2 / sqrt(D)`, pooled per level :20-21):
    d fmap1 = sum_l dvol_l . fmap2_l and d fmap2_l = dvol_l^T . fmap1 -- two launches of the library's own 128 x 128-tile matrix-core
    kernel (include/liso_slim.h: liso_corr_bwd_features_f32; all levels per launch, split-K partial sums added in a fixed order) in
    the arithmetic of the process's fp32 convolutions (F32X3 or exact fp32).  Until round 5: eight rocBLAS batched GEMMs per step.
    fmap1 [B, hw, D], levels[l] [B, H_l, W_l, D], dvol[l] [B, hw, H_l W_l] -> (g1 like fmap1, [g2_l like levels[l]])"""
    from liso_amd.utils import mfma_conv as MC

    L.require_cuda(fmap1, *levels, *dvol)
    B, hw, D = fmap1.shape
    h, w = levels[0].shape[1], levels[0].shape[2]
    assert h * w == hw and all(tuple(l.shape) == (B, h >> i, w >> i, D) for i, l in enumerate(levels)), [tuple(l.shape) for l in levels]
    assert all(tuple(dv.shape) == (B, hw, (h >> i) * (w >> i)) for i, dv in enumerate(dvol)), [tuple(dv.shape) for dv in dvol]
    cfg = L.CorrCfg(B, h, w, D, len(levels), 0)
    fmap1 = fmap1.contiguous()
    levels = [l.contiguous() for l in levels]
    dvol = [dv.contiguous() for dv in dvol]
    g1 = torch.empty_like(fmap1)
    g2 = [torch.empty_like(l) for l in levels]
    lib = L.lib()
    nbytes = lib.liso_corr_bwd_features_workspace_bytes(ctypes.byref(cfg))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=fmap1.device)
    mode = L.CONV_F32 if MC.fp32_mode() == "exact" else L.CONV_F32X3
    flops = 2 * 2.0 * B * hw * D * sum(dv.shape[2] for dv in dvol)
    with torch.cuda.device(fmap1.device):
        L.check(L.TIMER.launch("corr_bwd_features", lambda: lib.liso_corr_bwd_features_f32(
            ctypes.byref(cfg), mode, L.ptr(fmap1), _ptr_array(levels), _ptr_array(dvol), L.ptr(g1), _ptr_array(g2), L.ptr(ws), nbytes,
            L.stream_ptr()), units=flops), "corr_bwd_features")
    return g1, g2


class _Pyramid(torch.autograd.Function):
    """fmap2 [B, h, w, D] (channels last, = level 0) -> the pooled levels 1 .. L-1 in ONE launch (include/liso_slim.h:
    liso_corr_pyramid_fwd_f32; F.avg_pool2d per level until round 5); backward = one launch that folds the levels' gradients back onto
    level 0 (liso_corr_pyramid_bwd_f32)."""

    @staticmethod
    def forward(ctx, f2, num_levels):
        L.require_cuda(f2)
        B, h, w, D = f2.shape
        cfg = L.CorrCfg(B, h, w, D, num_levels, 0)
        outs = [torch.empty((B, h >> i, w >> i, D), dtype=torch.float32, device=f2.device) for i in range(1, num_levels)]
        with torch.cuda.device(f2.device):
            L.check(L.lib().liso_corr_pyramid_fwd_f32(ctypes.byref(cfg), L.ptr(f2), _ptr_array([f2] + outs), L.stream_ptr()), "corr_pyramid_fwd")
        ctx.cfg = cfg
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        cfg = ctx.cfg
        gs = [None] + [g.contiguous() if g is not None else None for g in grads]
        ref = next(g for g in gs if g is not None)
        out = torch.empty((cfg.batch, cfg.h, cfg.w, cfg.dim), dtype=torch.float32, device=ref.device)
        ptrs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() if g is not None else None for g in gs])
        with torch.cuda.device(ref.device):
            L.check(L.lib().liso_corr_pyramid_bwd_f32(ctypes.byref(cfg), ptrs, L.ptr(out), L.stream_ptr()), "corr_pyramid_bwd")
        return out, None


def lookup_forward(fmap1, fmap2_levels, coords, radius, out=None):
    """one correlation lookup (corr.py:23-46 of the reference) on detached features: fmap1 [B, hw, D], levels [B, H_i, W_i, D],
    coords [B, 2, h, w] (fp32, contiguous) -> out [B, h, w, L (2r+1)^2] (written into `out` when given).  -> (out, cfg)"""
    L.require_cuda(fmap1, coords)
    B, hw, D = fmap1.shape
    _, _, h, w = coords.shape
    cfg = L.CorrCfg(B, h, w, D, len(fmap2_levels), radius)
    W7 = 2 * radius + 1
    if out is None:
        out = torch.empty((B, h, w, len(fmap2_levels) * W7 * W7), dtype=torch.float32, device=fmap1.device)
    assert out.is_contiguous() and tuple(out.shape) == (B, h, w, len(fmap2_levels) * W7 * W7) and coords.is_contiguous()
    # the arithmetic of the process's fp32 convolutions: bf16 hi / lo pairs on the matrix cores ("x3": the tiled kernel, 4 x 8 queries
    # share the rows they correlate with) or fp32 FMAs ("exact": one wavefront per query and level)
    from liso_amd.utils import mfma_conv as MC

    tiled = MC.fp32_mode() == "x3" and os.environ.get("LISO_CORR_TILED", "1") != "0"
    fn = L.lib().liso_corr_lookup_fwd_tiled_f32 if tiled else L.lib().liso_corr_lookup_fwd_f32
    with torch.cuda.device(fmap1.device):
        L.check(L.TIMER.launch("corr_lookup_fwd_tiled" if tiled else "corr_lookup_fwd", lambda: fn(
            ctypes.byref(cfg), L.ptr(fmap1), _ptr_array(fmap2_levels), L.ptr(coords), L.ptr(out), L.stream_ptr())),
            "corr_lookup_fwd")
    return out, cfg


def lookup_backward(state, cfg, level_hw, coords, grad_out):
    """adds the adjoint of one lookup's bilinear windows to the CorrBlock's dense volume gradients `state.dvol` (created on first use);
    grad_out [B, h, w, C] fp32 contiguous"""
    g = grad_out
    assert g.is_contiguous() and g.dtype == torch.float32
    if state.dvol is None:
        state.dvol = [torch.zeros((cfg.batch, cfg.h * cfg.w, n), dtype=torch.float32, device=g.device) for n in level_hw]
    with torch.cuda.device(g.device):
        L.check(L.TIMER.launch("corr_lookup_bwd", lambda: L.lib().liso_corr_lookup_bwd_dvol_f32(
            ctypes.byref(cfg), L.ptr(coords), L.ptr(g), _ptr_array(state.dvol), L.stream_ptr())), "corr_lookup_bwd")


class _CorrLookup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coords, radius, state, token, fmap1, *fmap2_levels):
        coords = coords.detach().float().contiguous()
        out, cfg = lookup_forward(fmap1, fmap2_levels, coords, radius)
        ctx.save_for_backward(coords)
        ctx.cfg, ctx.state = cfg, state
        ctx.level_hw = [l.shape[1] * l.shape[2] for l in fmap2_levels]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (coords,) = ctx.saved_tensors
        cfg = ctx.cfg
        g = grad_out.float().contiguous()
        lookup_backward(ctx.state, cfg, ctx.level_hw, coords, g)
        return (None, None, None, g.new_zeros(1)) + (None,) * (1 + cfg.levels)


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        assert radius <= 3, "the gfx950 lookup kernel holds the (2r+2)^2 integer patch in one wavefront (r <= 3)"
        self.num_levels, self.radius = num_levels, radius
        B, D, h, w = fmap1.shape
        # channels-last query features [B, hw, D]
        self.fmap1 = fmap1.float().permute(0, 2, 3, 1).reshape(B, h * w, D).contiguous()
        # avg_pool2d of the correlation volume over (h2, w2) (reference :20-21) == correlation with the pooled fmap2
        f2 = fmap2.float().permute(0, 2, 3, 1).contiguous()  # (the encoders' maps are channels last: a view)
        self.levels = [f2] + (list(_Pyramid.apply(f2, num_levels)) if num_levels > 1 else [])
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
