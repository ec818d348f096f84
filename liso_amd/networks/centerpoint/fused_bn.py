"""Fused BatchNorm2d(+ReLU) on channels-last tensors through the gfx950 kernels of include/liso_bn.h.

`bn_act(x, bn, relu)` applies an nn.BatchNorm2d module's parameters/buffers (so state_dict keys stay those of the
reference's modules) with training/eval semantics identical to torch.nn.BatchNorm2d, but
  * 2 streaming launches forward, 2 backward (torch/MIOpen: 3 + 3 + separate ReLU kernels),
  * statistics merged with Chan's formula in fp64 (no E[x^2]-E[x]^2 cancellation),
  * fp32 or bf16 activations, fp32 parameters and statistics.
CPU tensors (host-logic unit tests only) take the plain torch path.
"""
import torch
import torch.nn.functional as F

from liso_amd import _lib as L


def _supported(c, dtype):
    v = 8 if dtype == torch.bfloat16 else 4
    return dtype in (torch.float32, torch.bfloat16) and c % v == 0 and c <= 256


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, relu):
        # x: logical [N,C,H,W] with channels-last storage
        N, C, H, W = x.shape
        xc = x.permute(0, 2, 3, 1)
        if not xc.is_contiguous():
            xc = xc.contiguous()
        M = N * H * W
        lib = L.lib()
        y = torch.empty_like(xc)
        stats = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        nbytes = lib.liso_bn_workspace_bytes(C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.TIMER.launch("bn_fwd", lambda: lib.liso_bn_relu_fwd(
                L.ptr(xc), int(x.dtype == torch.bfloat16), M, C, L.ptr(gamma), L.ptr(beta), L.ptr(running_mean),
                L.ptr(running_var), float(momentum), float(eps), int(training), int(relu), L.ptr(y), L.ptr(stats), L.ptr(ws),
                nbytes, L.stream_ptr()), units=3 * M * C * xc.element_size()), "bn_relu_fwd")
        ctx.save_for_backward(xc, gamma, stats)
        ctx.cfg = (M, C, bool(training), bool(relu))
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_y):
        xc, gamma, stats = ctx.saved_tensors
        M, C, training, relu = ctx.cfg
        g = grad_y.permute(0, 2, 3, 1)
        if g.dtype != xc.dtype:
            g = g.to(xc.dtype)
        if not g.is_contiguous():
            g = g.contiguous()
        lib = L.lib()
        dx = torch.empty_like(xc)
        gg = torch.empty(C, dtype=torch.float32, device=xc.device)
        gb = torch.empty(C, dtype=torch.float32, device=xc.device)
        nbytes = lib.liso_bn_workspace_bytes(C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xc.device)
        with torch.cuda.device(xc.device):
            L.check(L.TIMER.launch("bn_bwd", lambda: lib.liso_bn_relu_bwd(
                L.ptr(g), L.ptr(xc), int(xc.dtype == torch.bfloat16), M, C, L.ptr(gamma), L.ptr(stats), int(training),
                int(relu), L.ptr(dx), L.ptr(gg), L.ptr(gb), L.ptr(ws), nbytes, L.stream_ptr()),
                units=5 * M * C * xc.element_size()), "bn_relu_bwd")
        return dx.permute(0, 3, 1, 2), gg, gb, None, None, None, None, None, None


def bn_act(x, bn, relu=True):
    """y = ReLU?(BatchNorm2d(x)) with `bn`'s parameters; updates running stats / num_batches_tracked like the module."""
    training = bn.training or not bn.track_running_stats
    if bn.training and bn.track_running_stats and not getattr(bn, "_liso_counter_deferred", False):
        bn.num_batches_tracked += 1  # (a trainer may take over: one fused increment for all layers, see defer_batch_counters)
    if x.is_cuda and _supported(x.shape[1], x.dtype) and bn.affine and bn.track_running_stats:
        return _BnAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, training, relu)
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
    return F.relu(y, inplace=True) if relu else y


def defer_batch_counters(module):
    """Let the caller bump `num_batches_tracked` of every BatchNorm2d under `module` with ONE fused launch per step
    (`step_batch_counters`) instead of one tiny kernel per layer and forward (24 in the CenterPoint backbone)."""
    counters = []
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d) and m.track_running_stats:
            m._liso_counter_deferred = True
            counters.append(m.num_batches_tracked)
    return counters


def step_batch_counters(counters):
    if counters:
        torch._foreach_add_(counters, 1)
