"""symmetric_orthogonalization: R = U Vh of the SVD of a batch of 3x3 matrices, with the analytic backward.

Mirror of liso/torch_symm_ortho/__init__.py (same function name and autograd semantics).  The reference calls
torch.linalg.svd in fp64 (:63) and builds a [.., 3,3,3,3] derivative tensor in backward (:15-43); here both passes are
one small gfx950 kernel each (hand-written 3x3 one-sided Jacobi, closed-form backward grad_A = U (W - W^T) Vh),
include/liso_kabsch.h.  Inputs may be fp32 or fp64 with any leading batch shape; the solve itself is fp64.
"""
import torch

from liso_amd import _lib as L


class SymmetricOrthogonalization(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input):
        assert input.dtype in {torch.float, torch.double}, "only real floating point matrices"
        assert input.shape[-1] == 3 and input.shape[-2] == 3, "3x3 matrices required in the last two dimensions"
        L.require_cuda(input)
        a = input.detach().to(torch.double).contiguous()
        n = a.numel() // 9
        r, u, vh = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        d = torch.empty(a.shape[:-1], dtype=torch.double, device=a.device)
        with torch.cuda.device(a.device):
            L.check(L.lib().liso_symm_ortho_fwd_f64(L.ptr(a), n, L.ptr(r), L.ptr(u), L.ptr(vh), L.ptr(d), L.stream_ptr()),
                    "symm_ortho_fwd")
        ctx.save_for_backward(u, vh, d)
        ctx.in_dtype = input.dtype
        return r.to(input.dtype)

    @staticmethod
    def backward(ctx, grad_R):
        u, vh, d = ctx.saved_tensors
        g = grad_R.to(torch.double).contiguous()
        ga = torch.empty_like(g)
        with torch.cuda.device(g.device):
            L.check(L.lib().liso_symm_ortho_bwd_f64(L.ptr(g), L.ptr(u), L.ptr(vh), L.ptr(d), g.numel() // 9, L.ptr(ga),
                                                    L.stream_ptr()), "symm_ortho_bwd")
        return ga.to(ctx.in_dtype)


symmetric_orthogonalization = SymmetricOrthogonalization.apply
