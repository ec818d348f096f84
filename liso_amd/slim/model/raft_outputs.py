"""Network outputs of all RAFT iterations in one launch (liso_raft_upsample_outputs_*_f32, include/liso_slim.h): x8
bilinear upsampling of the low-resolution flow / logit maps + flow convention + concat + channels-last layout, i.e. what
raft_mod.py:244-258 of the reference does per iteration with upflow_n / uplogits_n / change_flow_convention_from_raft2usfl /
concat2network_output, and the adjoint of all of it in two launches."""
import ctypes

import torch

from liso_amd import _lib as L


class _RaftOutputs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow_lr, logits_lr, dirs, factor, flow_scale):
        L.require_cuda(flow_lr, logits_lr)
        n_it, b2, _, h, w = flow_lr.shape
        assert flow_lr.shape[2] == 2 and logits_lr.shape == (n_it, b2, 4, h, w), (flow_lr.shape, logits_lr.shape)
        f, lg = flow_lr.float().contiguous(), logits_lr.float().contiguous()
        cfg = L.UpsampleCfg(n_it, b2, dirs, h, w, factor, float(flow_scale))
        out = torch.empty((n_it * b2, h * factor, w * factor, 8), dtype=torch.float32, device=f.device)
        with torch.cuda.device(f.device):
            L.check(L.TIMER.launch("raft_outputs_fwd", lambda: L.lib().liso_raft_upsample_outputs_fwd_f32(
                ctypes.byref(cfg), L.ptr(f), L.ptr(lg), L.ptr(out), L.stream_ptr())), "raft_outputs_fwd")
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, grad_out):
        cfg = ctx.cfg
        g = grad_out.float().contiguous()
        nbytes = int(L.lib().liso_raft_upsample_scratch_bytes(ctypes.byref(cfg)))
        scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=g.device)
        gf = torch.empty((cfg.n_it, cfg.batch2, 2, cfg.h, cfg.w), dtype=torch.float32, device=g.device)
        gl = torch.empty((cfg.n_it, cfg.batch2, 4, cfg.h, cfg.w), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            L.check(L.TIMER.launch("raft_outputs_bwd", lambda: L.lib().liso_raft_upsample_outputs_bwd_f32(
                ctypes.byref(cfg), L.ptr(g), L.ptr(scratch), nbytes, L.ptr(gf), L.ptr(gl), L.stream_ptr())), "raft_outputs_bwd")
        return gf, gl, None, None, None


def raft_network_outputs(flows_lr, logits_lr, *, dirs, factor, resolution_adapter):
    """flows_lr / logits_lr: lists (one entry per RAFT iteration) of [2B,2,h,w] (coords1 - coords0) and [2B,4,h,w] ->
    [n_it*2B, H, W, 8] network outputs ordered [dir][iteration][sample] (see include/liso_slim.h)."""
    return _RaftOutputs.apply(torch.stack(flows_lr, dim=0), torch.stack(logits_lr, dim=0), dirs, factor,
                              factor * float(resolution_adapter))  # n * interpolate(flow), then * adapter (raft_mod.py:262-266)
