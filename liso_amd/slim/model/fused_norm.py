"""InstanceNorm2d(+ReLU) of the SLIM encoders in training, on channels-last tensors through the grouped passes of include/liso_bn.h.

`in_act(x, norm, relu)` applies an `nn.InstanceNorm2d` module (liso/slim/model/extractor.py:24-38,219-230: norm_fn "instance" /
"instance_affine", no running statistics) with its own parameters, forward and backward: 3 + 3 launches on the layout the
convolutions produce and consume -- PyTorch routes InstanceNorm through MIOpen's BatchNorm on a [1, B*C, H, W] view in NCHW
(weight / bias repeated per sample, a layout copy in front of the next convolution, a separate ReLU and its mask pass backward).
Statistics are merged in fp64 from block-shifted sums (no E[x^2] - E[x]^2 cancellation).  Layers this does not cover (CPU tensors,
running statistics, odd channel counts) take the module itself.
"""
import torch
import torch.nn.functional as F

from liso_amd import _lib as L

_CONST = {}


def _const(value, c, device):
    key = (value, c, str(device))
    if key not in _CONST:
        _CONST[key] = torch.full((c,), value, dtype=torch.float32, device=device)
    return _CONST[key]


def supported(x, norm):
    v = 8 if x.dtype == torch.bfloat16 else 4
    c = x.shape[1]
    return (isinstance(norm, torch.nn.InstanceNorm2d) and not norm.track_running_stats and x.is_cuda and x.dim() == 4
            and x.dtype in (torch.float32, torch.bfloat16) and c % v == 0 and c <= 256 and x.shape[2] * x.shape[3] > 0)


class _InAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu):
        B, C, H, W = x.shape  # logical [B,C,H,W], channels-last storage
        xc = x.permute(0, 2, 3, 1)
        if not xc.is_contiguous():
            xc = xc.contiguous()
        lib = L.lib()
        y = torch.empty_like(xc)
        stats = torch.empty((B, 4 * C), dtype=torch.float32, device=x.device)
        nbytes = lib.liso_in_workspace_bytes(B, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.TIMER.launch("in_fwd", lambda: lib.liso_in_relu_fwd(
                L.ptr(xc), int(x.dtype == torch.bfloat16), B, H * W, C, L.ptr(gamma), L.ptr(beta), float(eps), int(relu), L.ptr(y),
                L.ptr(stats), L.ptr(ws), nbytes, L.stream_ptr()), units=3 * xc.numel() * xc.element_size()), "in_relu_fwd")
        ctx.save_for_backward(xc, gamma, stats)
        ctx.relu = bool(relu)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_y):
        xc, gamma, stats = ctx.saved_tensors
        B, H, W, C = xc.shape
        g = grad_y.permute(0, 2, 3, 1)
        if g.dtype != xc.dtype:
            g = g.to(xc.dtype)
        if not g.is_contiguous():
            g = g.contiguous()
        lib = L.lib()
        dx = torch.empty_like(xc)
        gg = torch.empty(C, dtype=torch.float32, device=xc.device)  # (summed over the samples inside the finalize launch)
        gb = torch.empty(C, dtype=torch.float32, device=xc.device)
        nbytes = lib.liso_in_workspace_bytes(B, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xc.device)
        with torch.cuda.device(xc.device):
            L.check(L.TIMER.launch("in_bwd", lambda: lib.liso_in_relu_bwd_sum(
                L.ptr(g), L.ptr(xc), int(xc.dtype == torch.bfloat16), B, H * W, C, L.ptr(gamma), L.ptr(stats), int(ctx.relu), L.ptr(dx),
                L.ptr(gg), L.ptr(gb), L.ptr(ws), nbytes, L.stream_ptr()), units=5 * xc.numel() * xc.element_size()), "in_relu_bwd")
        need = ctx.needs_input_grad
        return dx.permute(0, 3, 1, 2), (gg if need[1] else None), (gb if need[2] else None), None, None


def in_act(x, norm, relu=True):
    """y = ReLU?(norm(x)) for an nn.InstanceNorm2d `norm`"""
    if not supported(x, norm):
        y = norm(x)
        return F.relu(y, inplace=True) if relu else y
    c = x.shape[1]
    gamma = norm.weight if norm.affine else _const(1.0, c, x.device)
    beta = norm.bias if norm.affine else _const(0.0, c, x.device)
    return _InAct.apply(x, gamma, beta, norm.eps, relu)
