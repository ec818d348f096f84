"""GPU parity: fused BatchNorm2d(+ReLU) kernels vs an fp64 torch reference of the same op (tolerance 1e-3 relative;
they are in fact accurate to ~1e-6 where MIOpen's spatial BN loses 1e-4..1e-2 on inputs with large mean/std)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def _ref(x, bn_w, bn_b, rm, rv, training, relu, eps, momentum):
    y = F.batch_norm(x, rm, rv, bn_w, bn_b, training, momentum, eps)
    return F.relu(y) if relu else y


@pytest.mark.parametrize("C,HW,N,offset,relu,training", [(64, 64, 2, 0.0, True, True), (128, 32, 3, 5.0, True, True),
                                                          (256, 16, 2, 50.0, False, True), (64, 128, 4, 0.0, True, False),
                                                          (128, 20, 1, -3.0, True, True)])
def test_bn_relu_fp32_matches_fp64_reference(C, HW, N, offset, relu, training):
    from liso_amd.networks.centerpoint.fused_bn import bn_act

    torch.manual_seed(C + HW)
    x = (torch.randn(N, C, HW, HW) * 0.1 + offset).cuda().contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
    bn.train(training)
    rm0, rv0 = bn.running_mean.clone().double(), bn.running_var.clone().double()
    xd = x.detach().double().requires_grad_(True)
    wd, bd = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    yd = _ref(xd, wd, bd, rm0, rv0, training, relu, bn.eps, bn.momentum)
    g = torch.randn_like(yd)
    (yd * g).sum().backward()
    xg = x.clone().requires_grad_(True)
    y = bn_act(xg, bn, relu=relu)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    (y * g.float()).sum().backward()
    tol_f = 1e-5 if offset < 10 else 2e-3   # forward conditioning degrades as |mean|/std grows (fp32 input rounding)
    assert _rel(y, yd) < tol_f
    if training:
        assert _rel(bn.running_mean, rm0) < 1e-5 and _rel(bn.running_var, rv0) < 1e-4
        assert int(bn.num_batches_tracked) == 1
    # backward: ill-conditioned for large offsets in ANY fp32 implementation; compare where it is well posed
    if abs(offset) <= 5:
        tol_b = 1e-3 if offset == 0 else 5e-2
        assert _rel(xg.grad, xd.grad) < tol_b
        assert _rel(bn.weight.grad, wd.grad) < tol_b and _rel(bn.bias.grad, bd.grad) < 1e-4


def test_bn_relu_bf16_and_better_than_library_bn():
    from liso_amd.networks.centerpoint.fused_bn import bn_act

    torch.manual_seed(0)
    x = (torch.randn(2, 64, 64, 64) * 0.1 + 5).cuda().contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(64).cuda().train()
    ref = F.relu(F.batch_norm(x.double(), None, None, bn.weight.double(), bn.bias.double(), True, 0.1, bn.eps))
    ours = bn_act(x, bn, relu=True)
    lib = F.relu(F.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.1, bn.eps))
    assert _rel(ours, ref) < 1e-5
    assert _rel(ours, ref) <= _rel(lib, ref)  # never worse than the library BN it replaces
    xb = x.bfloat16()
    yb = bn_act(xb, bn, relu=True)
    refb = F.relu(F.batch_norm(xb.double(), None, None, bn.weight.double(), bn.bias.double(), True, 0.1, bn.eps))
    assert yb.dtype == torch.bfloat16 and _rel(yb, refb) < 1e-2
    g = torch.randn_like(yb)
    xb2 = xb.clone().requires_grad_(True)
    (bn_act(xb2, bn, relu=True) * g).sum().backward()
    assert torch.isfinite(xb2.grad.float()).all() and xb2.grad.dtype == torch.bfloat16


@pytest.mark.parametrize("C,H,W,B,affine,relu,dtype", [(32, 64, 64, 2, True, True, torch.float32), (96, 16, 24, 3, True, True, torch.float32),
                                                       (64, 32, 32, 1, False, False, torch.float32), (96, 32, 32, 2, True, True, torch.bfloat16),
                                                       (24, 8, 8, 2, True, True, torch.float32)])
def test_instance_norm_relu_matches_fp64_module(C, H, W, B, affine, relu, dtype):
    """in_act (the grouped passes of include/liso_bn.h) vs nn.InstanceNorm2d (+ReLU) in fp64: the SLIM encoders' training norm
    (liso/slim/model/extractor.py:24-38), incl. 96 channels (24 lanes per row: does not divide the 256-thread block) and a bf16 tensor"""
    from liso_amd.slim.model.fused_norm import in_act, supported

    torch.manual_seed(C + H)
    x = (torch.randn(B, C, H, W) * torch.linspace(0.2, 2.0, C)[None, :, None, None] + 0.7).cuda().to(dtype)
    x = x.contiguous(memory_format=torch.channels_last)
    norm = torch.nn.InstanceNorm2d(C, eps=1e-3, affine=affine).cuda()
    if affine:
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5); norm.bias.uniform_(-0.5, 0.5)
    assert supported(x, norm)
    ref = torch.nn.InstanceNorm2d(C, eps=1e-3, affine=affine).cuda().double()
    if affine:
        ref.load_state_dict({k: v.double() for k, v in norm.state_dict().items()})
    xd = x.detach().double().requires_grad_(True)
    yd = ref(xd)
    yd = F.relu(yd) if relu else yd
    g = torch.randn_like(yd)
    (yd * g).sum().backward()
    xg = x.clone().requires_grad_(True)
    y = in_act(xg, norm, relu=relu)
    assert y.shape == x.shape and y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    (y.float() * g.float()).sum().backward()
    tf, tb = (1e-5, 2e-4) if dtype == torch.float32 else (1e-2, 3e-2)
    assert _rel(y, yd) < tf
    assert _rel(xg.grad, xd.grad) < tb
    if affine:
        assert _rel(norm.weight.grad, ref.weight.grad) < tb and _rel(norm.bias.grad, ref.bias.grad) < tb


@pytest.mark.parametrize("C,HW,N,dtype", [(64, 256, 2, torch.bfloat16), (128, 64, 4, torch.float32), (256, 32, 1, torch.bfloat16),
                                          (64, 8, 1, torch.float32)])
def test_bn_backward_with_last_block_finalize_equals_three_launch_form(C, HW, N, dtype, monkeypatch):
    """liso_bn_relu_bwd_ticket (the reduction's last block finalises: 2 launches) against liso_bn_relu_bwd (3 launches) on the same
    tensors: same dx / dgamma / dbeta (fp64 merge of the same partial sums, different chunking), the layer's counter is zero again
    after every call, and repeated calls reuse it"""
    from liso_amd.utils import mfma_conv as MC

    torch.manual_seed(C + HW)
    x = (torch.randn(N, C, HW, HW, device="cuda") * 0.5 + 0.2).to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(N, C, HW, HW, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    gamma = torch.nn.Parameter(torch.rand(C, device="cuda") + 0.5)
    mean = x.float().mean(dim=(0, 2, 3))
    invstd = (x.float().var(dim=(0, 2, 3), unbiased=False) + 1e-3).rsqrt()
    beta = torch.randn(C, device="cuda") * 0.1
    stats = torch.cat([gamma.detach() * invstd, beta - mean * gamma.detach() * invstd, mean, invstd]).contiguous()
    grp = {"gamma": gamma, "beta": None, "stats": stats}
    monkeypatch.setenv("LISO_BN_TICKET", "0")
    dx0, gg0, gb0 = MC._bn_backward_group(g, x, grp, True, True)
    monkeypatch.setenv("LISO_BN_TICKET", "1")
    for _ in range(3):
        dx1, gg1, gb1 = MC._bn_backward_group(g, x, grp, True, True)
        pool, slots = MC._BN_TICKETS[x.device.index]
        assert id(gamma) in slots and int(pool.abs().sum()) == 0
        assert _rel(gg1, gg0) < 1e-6 and _rel(gb1, gb0) < 1e-6
        assert _rel(dx1.float(), dx0.float()) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_bn_backward_on_channel_slices_equals_the_dense_call(dtype):
    """liso_bn_relu_bwd_strided: the BatchNorms behind a channel concatenation (3 x 128 channels in front of the detector's head) read
    their slices of g / x and write their slice of dx in place: same dx / dgamma / dbeta as the call on contiguous copies of the slices,
    bit for bit (same kernels, other row strides), and nothing outside the slice is touched"""
    from liso_amd.utils import mfma_conv as MC

    torch.manual_seed(3)
    B, H, W, Cs = 2, 32, 48, [128, 64, 128]
    Ct = sum(Cs)
    x = (torch.randn(B, H, W, Ct, device="cuda") * 0.5 + 0.1).to(dtype).permute(0, 3, 1, 2)      # logical NCHW, channels-last storage
    g = torch.randn(B, H, W, Ct, device="cuda").to(dtype).permute(0, 3, 1, 2)
    dx_full = torch.full((B, H, W, Ct), float("nan"), device="cuda").to(dtype).permute(0, 3, 1, 2)
    a = 0
    for C in Cs:
        gamma = torch.nn.Parameter(torch.rand(C, device="cuda") + 0.5)
        xs = x[:, a:a + C].float()
        mean, invstd = xs.mean(dim=(0, 2, 3)), (xs.var(dim=(0, 2, 3), unbiased=False) + 1e-3).rsqrt()
        beta = torch.randn(C, device="cuda") * 0.1
        grp = {"gamma": gamma, "beta": None, "stats": torch.cat([gamma.detach() * invstd, beta - mean * gamma.detach() * invstd, mean, invstd]).contiguous()}
        dx_s, gg_s, gb_s = MC._bn_backward_group(g[:, a:a + C], x[:, a:a + C], grp, True, True, out=dx_full[:, a:a + C])
        dx_d, gg_d, gb_d = MC._bn_backward_group(g[:, a:a + C].contiguous(memory_format=torch.channels_last),
                                                 x[:, a:a + C].contiguous(memory_format=torch.channels_last), grp, True, True)
        assert dx_s.data_ptr() == dx_full[:, a:a + C].data_ptr()  # (written in place)
        assert torch.equal(dx_s, dx_d) and torch.equal(gg_s, gg_d) and torch.equal(gb_s, gb_d)
        a += C
    assert not bool(torch.isnan(dx_full).any())  # every slice was written ...
    # ... and only its own channels: the neighbours of the middle group are intact after it is recomputed
    before = dx_full.clone()
    C0, C1 = Cs[0], Cs[1]
    gamma = torch.nn.Parameter(torch.ones(C1, device="cuda"))
    grp = {"gamma": gamma, "beta": None, "stats": torch.cat([torch.ones(C1), torch.zeros(C1), torch.zeros(C1), torch.ones(C1)]).cuda()}
    MC._bn_backward_group(g[:, C0:C0 + C1], x[:, C0:C0 + C1], grp, False, True, out=dx_full[:, C0:C0 + C1])
    assert torch.equal(dx_full[:, :C0], before[:, :C0]) and torch.equal(dx_full[:, C0 + C1:], before[:, C0 + C1:])
