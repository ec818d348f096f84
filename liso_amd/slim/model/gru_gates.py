"""Fused ConvGRU gate arithmetic (liso_gru_*_f32, include/liso_slim.h): the elementwise part of ConvGRU.forward
(liso/slim/model/update.py:29-37) between its convolutions, forward and backward."""
import ctypes

import torch

from liso_amd import _lib as L


def _cl(*ts):
    """all maps channels-last (the own convolutions' layout)?  Then a PIXEL plays the role of the kernels' `batch` entry with
    hw = 1: [pixel][channels] rows are exactly the [batch][channels * hw] rows the NCHW kernels walk."""
    return all(t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous() for t in ts)


def _cfg(h, x, cl=False):
    B, ch, H, W = h.shape
    if cl:
        return L.GruCfg(B * H * W, ch, x.shape[1], 1)
    return L.GruCfg(B, ch, x.shape[1], H * W)


class _GruIn(torch.autograd.Function):
    """zr [B,2ch,H,W] (pre-activations of z | r), h [B,ch,H,W], x [B,cx,H,W] -> z = sigmoid(zr[:, :ch]),
    rhx = cat([sigmoid(zr[:, ch:]) * h, x])"""

    @staticmethod
    def forward(ctx, zr, h, x):
        L.require_cuda(zr, h, x)
        cl = _cl(zr) or _cl(h)  # (the convolution's output decides; small NCHW operands are converted)
        fmt = torch.channels_last if cl else torch.contiguous_format
        zr, h, x = (t.float().contiguous(memory_format=fmt) for t in (zr, h, x))
        cfg = _cfg(h, x, cl)
        assert zr.shape[1] == 2 * cfg.ch, zr.shape
        z = torch.empty_like(h)
        rhx = torch.empty((h.shape[0], cfg.ch + cfg.cx) + tuple(h.shape[2:]), dtype=torch.float32, device=h.device, memory_format=fmt)
        bs = 2 * cfg.ch * cfg.hw
        with torch.cuda.device(h.device):
            L.check(L.lib().liso_gru_in_fwd_f32(ctypes.byref(cfg), L.ptr(zr), ctypes.c_void_p(zr.data_ptr() + 4 * cfg.ch * cfg.hw), bs,
                                                L.ptr(h), L.ptr(x), L.ptr(z), L.ptr(rhx), L.stream_ptr()), "gru_in_fwd")
        ctx.save_for_backward(zr, h, z)
        ctx.cfg, ctx.fmt = cfg, fmt
        ctx.set_materialize_grads(False)
        return z, rhx

    @staticmethod
    def backward(ctx, g_z, g_rhx):
        zr, h, z = ctx.saved_tensors
        cfg, fmt = ctx.cfg, ctx.fmt
        if g_rhx is None:
            g_rhx = torch.zeros((h.shape[0], cfg.ch + cfg.cx) + tuple(h.shape[2:]), dtype=torch.float32, device=h.device)
        g_rhx = g_rhx.float().contiguous(memory_format=fmt)
        gz = None if g_z is None else g_z.float().contiguous(memory_format=fmt)
        g_zr, g_h = torch.empty_like(zr), torch.empty_like(h)
        bs = 2 * cfg.ch * cfg.hw
        with torch.cuda.device(h.device):
            L.check(L.lib().liso_gru_in_bwd_f32(ctypes.byref(cfg), ctypes.c_void_p(zr.data_ptr() + 4 * cfg.ch * cfg.hw), bs, L.ptr(h),
                                                L.ptr(z), L.ptr(gz) if gz is not None else None, L.ptr(g_rhx), L.ptr(g_zr),
                                                ctypes.c_void_p(g_zr.data_ptr() + 4 * cfg.ch * cfg.hw), bs, L.ptr(g_h),
                                                L.stream_ptr()), "gru_in_bwd")
        return g_zr, g_h, g_rhx[:, cfg.ch:]


class _GruOut(torch.autograd.Function):
    """h' = (1 - z) * h + z * tanh(cq)"""

    @staticmethod
    def forward(ctx, cq, z, h):
        L.require_cuda(cq, z, h)
        fmt = torch.channels_last if (_cl(cq) or _cl(z)) else torch.contiguous_format  # elementwise: any COMMON layout works
        cq, z, h = (t.float().contiguous(memory_format=fmt) for t in (cq, z, h))
        ctx.fmt = fmt
        out = torch.empty_like(h)
        with torch.cuda.device(h.device):
            L.check(L.lib().liso_gru_out_fwd_f32(h.numel(), L.ptr(cq), L.ptr(z), L.ptr(h), L.ptr(out), L.stream_ptr()), "gru_out_fwd")
        ctx.save_for_backward(cq, z, h)
        return out

    @staticmethod
    def backward(ctx, g):
        cq, z, h = ctx.saved_tensors
        g = g.float().contiguous(memory_format=ctx.fmt)
        g_cq, g_z, g_h = torch.empty_like(cq), torch.empty_like(z), torch.empty_like(h)
        with torch.cuda.device(h.device):
            L.check(L.lib().liso_gru_out_bwd_f32(h.numel(), L.ptr(cq), L.ptr(z), L.ptr(h), L.ptr(g), L.ptr(g_cq), L.ptr(g_z), L.ptr(g_h),
                                                 L.stream_ptr()), "gru_out_bwd")
        return g_cq, g_z, g_h


gru_in = _GruIn.apply
gru_out = _GruOut.apply
