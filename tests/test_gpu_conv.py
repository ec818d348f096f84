"""GPU parity of the own MFMA convolutions (include/liso_conv.h) against torch's convolution evaluated in fp64 on the
same (rounded) operands: forward with the fused prologue / epilogue, data gradient, weight gradient, both arithmetic
modes (bf16; fp32 as bf16 hi/lo pairs "F32X3"; exact fp32 on the native fp32 MFMA), every geometry the BEV networks use (rpn.py:113-146, center_head.py:60-117,
update.py:96-164, extractor.py:211-297) plus ragged sizes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, Ci, Co, H, W, k, stride, pad, transposed)
GEOMS = [
    (2, 64, 64, 64, 64, 3, 1, 1, False),      # backbone 3x3
    (1, 64, 64, 40, 72, 3, 2, 1, False),      # stride-2 stage entry
    (2, 128, 128, 32, 32, 3, 1, 1, False),
    (1, 256, 256, 16, 32, 3, 1, 1, False),
    (2, 64, 128, 32, 64, 2, 2, 0, False),     # deblock 0: conv k2 s2
    (2, 128, 128, 32, 32, 1, 1, 0, False),    # deblock 1: conv 1x1
    (2, 256, 128, 16, 16, 2, 2, 0, True),     # deblock 2: transposed k2 s2
    (1, 384, 64, 32, 32, 3, 1, 1, False),     # head shared conv
    (2, 64, 3, 32, 32, 3, 1, 1, False),       # head output convs
    (2, 64, 1, 32, 32, 3, 1, 1, False),
    (1, 16, 32, 19, 45, 3, 1, 1, False),      # ragged map, partial tiles
    (3, 32, 96, 9, 7, 3, 2, 1, False),
    (1, 64, 32, 64, 64, 7, 2, 3, False),      # encoder stem 7x7 s2
    (2, 8, 64, 32, 32, 7, 1, 3, False),       # motion encoder 7x7 (channels padded to 8)
    (2, 200, 96, 16, 16, 1, 1, 0, False),     # correlation 1x1 (196 -> padded)
    (2, 400, 192, 16, 16, 3, 1, 1, False),    # ConvGRU z|r
    (1, 96, 64, 24, 24, 1, 2, 0, False),      # residual downsample 1x1 s2
]


def _ref_conv(x, w, b, s, p, transposed):
    x, w = x.double(), w.double()
    b = b.double() if b is not None else None
    return F.conv_transpose2d(x, w, b, stride=s, padding=p) if transposed else F.conv2d(x, w, b, stride=s, padding=p)


def _mk(geom, dtype, seed=0):
    B, Ci, Co, H, W, k, s, p, tr = geom
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    if dtype == torch.bfloat16:  # the kernel sees bf16 activations and bf16-rounded weights
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    return x, w, b


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


ARITH = [pytest.param((torch.bfloat16, "x3"), id="bf16"), pytest.param((torch.float32, "x3"), id="f32x3"),
         pytest.param((torch.float32, "exact"), id="f32exact")]


@pytest.fixture(autouse=True)
def _restore_fp32_mode():
    from liso_amd.utils import mfma_conv as MC

    prev = MC.fp32_mode()
    yield
    MC.set_fp32_mode(prev)


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("arith", ARITH)
def test_forward_dgrad_wgrad(geom, arith, monkeypatch):
    from liso_amd.utils import mfma_conv as MC

    dtype, fmode = arith
    MC.set_fp32_mode(fmode)
    exact = dtype == torch.float32 and fmode == "exact"
    B, Ci, Co, H, W, k, s, p, tr = geom
    x, w, b = _mk(geom, dtype)
    spec = MC.ConvSpec(k, k, s, p, tr)
    xd = x.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    wd, bd = w.cuda(), b.cuda()
    y, _ = MC.conv_forward(xd, wd, bd, spec, out_dtype=torch.float32)
    ref = _ref_conv(x, w, b, s, p, tr)
    assert y.shape == ref.shape
    # bf16 mode: exact products, fp32 accumulation; F32X3: 2^-16 per product; exact: 2^-24 per product, fp32 accumulation
    tol = 2e-5 if dtype == torch.bfloat16 else 3e-6 if exact else 6e-5
    assert _rel(y, ref) <= tol, _rel(y, ref)
    if dtype == torch.bfloat16:  # bf16 output = correctly rounded fp32 result (ties aside)
        y16, _ = MC.conv_forward(xd, wd, bd, spec)
        assert y16.dtype == torch.bfloat16 and _rel(y16, ref) <= 5e-3
    # data gradient and weight gradient against autograd in fp64
    g = torch.Generator().manual_seed(7)
    dy = torch.randn(ref.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    out = _ref_conv(x64, w64, b64, s, p, tr)
    gx, gw, gb = torch.autograd.grad(out, [x64, w64, b64], dy.double())
    dyd = dy.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    dx = MC.conv_dgrad(dyd, wd, spec, tuple(x.shape), out_dtype=torch.float32)
    assert _rel(dx, gx) <= tol, _rel(dx, gx)
    # (dense 64-channel 7x7 stem: the row-of-taps MFMA kernel -- the product path feeds an occupancy map and takes the sparse kernels)
    res = MC.conv_wgrad(xd, dyd, tuple(w.shape), spec)
    assert res is not None  # every geometry of the networks can run on the own kernel (7 x 7: one kernel row of taps per block)
    dw, db = res
    assert _rel(dw, gw) <= 2 * tol, _rel(dw, gw)
    assert _rel(db, gb) <= 2 * tol, _rel(db, gb)


@pytest.mark.parametrize("arith", ARITH)
def test_prologue_epilogue_statistics_and_channel_slices(arith):
    """x' = relu(x * scale + shift) fused into the staging (padding stays exactly zero), ReLU + bias epilogue, the partial
    sums -> BatchNorm statistics (vs torch on the stored tensor), input given as a channel slice of a wider tensor"""
    from liso_amd.utils import mfma_conv as MC

    dtype, fmode = arith
    MC.set_fp32_mode(fmode)
    exact = dtype == torch.float32 and fmode == "exact"
    geom = (2, 64, 128, 40, 40, 3, 1, 1, False)
    B, Ci, Co, H, W, k, s, p, tr = geom
    x, w, b = _mk(geom, dtype, seed=3)
    g = torch.Generator().manual_seed(5)
    scale, shift = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.5
    wide = torch.randn(B, 3 * Ci, H, W, generator=g)
    wide[:, Ci:2 * Ci] = x
    wide_d = wide.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    xs = wide_d[:, Ci:2 * Ci]  # a view: pixel stride 3 * Ci
    spec = MC.ConvSpec(k, k, s, p, tr)
    xin = torch.relu(x.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    if dtype == torch.bfloat16:
        xin = torch.relu((x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))).bfloat16().double()  # rounded like the staging does
    ref = torch.relu(F.conv2d(xin, w.double(), b.double(), stride=s, padding=p))
    bn = torch.nn.BatchNorm2d(Co).cuda().train()
    with torch.no_grad():
        bn.running_mean.uniform_(-0.2, 0.2)
    shift_stat = bn.running_mean.clone()
    y, part = MC.conv_forward(xs, w.cuda(), b.cuda(), spec, scale.cuda(), shift.cuda(), in_relu=True, out_relu=True,
                              want_stats=True, stats_shift=shift_stat)
    tol = 8e-3 if dtype == torch.bfloat16 else 3e-6 if exact else 6e-5
    assert _rel(y, ref) <= tol, _rel(y, ref)
    fold = MC.finalize_bn(part, B * y.shape[2] * y.shape[3], bn, shift_stat)
    yf = y.float()
    mean, var = yf.mean(dim=(0, 2, 3)), yf.var(dim=(0, 2, 3), unbiased=False)
    st = fold.groups[0]["stats"]
    assert _rel(st[2 * Co:3 * Co], mean) <= 1e-4
    assert _rel(st[3 * Co:], torch.rsqrt(var + bn.eps)) <= 1e-4
    assert _rel(st[:Co], bn.weight * torch.rsqrt(var + bn.eps)) <= 1e-4
    n = B * y.shape[2] * y.shape[3]
    assert _rel(bn.running_mean, 0.9 * shift_stat + 0.1 * mean) <= 1e-4
    assert _rel(bn.running_var, 0.9 * torch.ones_like(var) + 0.1 * var * n / (n - 1)) <= 1e-4
    # weight gradient with the same prologue
    dy = torch.randn(ref.shape, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    w64 = w.double().requires_grad_(True)
    gw, = torch.autograd.grad(F.conv2d(xin, w64, None, stride=s, padding=p), [w64], dy.double())
    dw, _ = MC.conv_wgrad(xs, dy.to(dtype).cuda().contiguous(memory_format=torch.channels_last), tuple(w.shape), spec,
                          scale.cuda(), shift.cuda(), in_relu=True)
    assert _rel(dw, gw) <= (2e-4 if dtype == torch.bfloat16 else 6e-6 if exact else 1e-4), _rel(dw, gw)


def _stat(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).abs()
    return float(d.max() / b.abs().max()), float(d.median() / b.abs().median().clamp(min=1e-30)), \
        float((d > 1e-3 * b.abs().max()).double().mean())


@pytest.mark.parametrize("fmode", ["x3", "exact"])
@pytest.mark.parametrize("beta_lo,beta_hi", [(3.0, 4.0), (-0.3, 0.3)])
def test_fused_conv_autograd_chain_matches_torch_modules(beta_lo, beta_hi, fmode):
    """conv -> BN -> ReLU -> conv -> BN -> ReLU -> conv(bias) as the networks chain them (BnFold), training mode, fp32
    tensors: outputs, input gradient, every parameter gradient and the running statistics against torch.nn modules in fp64.
    With beta in [3, 4] every ReLU is open, the chain is smooth and everything must agree tightly.  With beta around 0 a
    ReLU whose input is within the forward rounding error (1e-5 in F32X3 mode) of zero may open on one side only -- the
    gradient of that single pixel then differs, and through the BatchNorm statistics every gradient upstream moves a
    little -- so the bulk of every gradient is checked: median error, fraction of outliers."""
    import copy

    from liso_amd.utils import mfma_conv as MC

    MC.set_fp32_mode(fmode)
    torch.manual_seed(0)
    convs = [torch.nn.Conv2d(32, 64, 3, stride=2, padding=1, bias=False), torch.nn.Conv2d(64, 64, 3, padding=1, bias=True),
             torch.nn.Conv2d(64, 8, 3, padding=1, bias=True)]
    bns = [torch.nn.BatchNorm2d(64), torch.nn.BatchNorm2d(64)]
    with torch.no_grad():
        for bn in bns:
            bn.weight.uniform_(0.5, 1.5), bn.bias.uniform_(beta_lo, beta_hi)
    x = torch.randn(2, 32, 48, 40)
    ref = copy.deepcopy(torch.nn.Sequential(convs[0], bns[0], torch.nn.ReLU(), convs[1], bns[1], torch.nn.ReLU(), convs[2])).double().train()
    x64 = x.double().requires_grad_(True)
    out64 = ref(x64)
    wgt = torch.linspace(-1, 1, out64.numel()).view_as(out64).double()
    (out64 * wgt).sum().backward()
    for m in convs + bns:
        m.cuda().train()
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y, fold = MC.fused_conv(xd, None, convs[0], out_bn=bns[0])
    y, fold = MC.fused_conv(y, fold, convs[1], out_bn=bns[1])
    y, _ = MC.fused_conv(y, fold, convs[2])
    assert _rel(y, out64) <= (5e-6 if fmode == "exact" else 1e-4), _rel(y, out64)
    (y * wgt.float().cuda()).sum().backward()
    smooth = beta_lo > 1.0
    pairs = [("x", xd.grad, x64.grad)] + [(f"conv{i}.w", convs[i].weight.grad, ref[j].weight.grad) for i, j in ((0, 0), (1, 3), (2, 6))] + \
        [("conv2.b", convs[2].bias.grad, ref[6].bias.grad)] + \
        [(f"bn{i}.{n}", getattr(bns[i], n).grad, getattr(ref[j], n).grad) for i, j in ((0, 1), (1, 4)) for n in ("weight", "bias")]
    for name, a, b in pairs:
        mx, med, frac = _stat(a, b)
        if smooth:
            assert mx <= (2e-5 if fmode == "exact" else 1e-3), (name, mx, med)
        elif fmode == "exact":  # a kink flips only within fp32 rounding of zero: the pre-F32X3 limits
            assert med <= 1e-3 and mx <= 4e-3, (name, mx, med, frac)
        else:
            assert med <= 1e-2 and mx <= 0.5, (name, mx, med, frac)
    for i, j in ((0, 1), (1, 4)):
        assert _rel(bns[i].running_mean, ref[j].running_mean) <= 1e-4 and _rel(bns[i].running_var, ref[j].running_var) <= 1e-4
        assert int(bns[i].num_batches_tracked) == 1


@pytest.mark.parametrize("arith", ARITH[:2])
def test_sparse_input_tile_skipping_is_bit_identical(arith):
    """liso_conv_forward_sparse: with the occupancy map of a sparse canvas (3 % of the cells occupied, clustered) the blocks whose
    input window holds no occupied cell skip their work; output and BatchNorm partial sums must equal the dense call bit for bit"""
    from liso_amd.utils import mfma_conv as MC

    dtype, fmode = arith
    MC.set_fp32_mode(fmode)
    g = torch.Generator().manual_seed(3)
    B, C, H, W = 2, 64, 256, 256
    occ = torch.zeros(B, 1, H, W)
    for b in range(B):
        for _ in range(12):
            y0, x0 = int(torch.randint(0, H - 24, (1,), generator=g)), int(torch.randint(0, W - 24, (1,), generator=g))
            occ[b, 0, y0:y0 + 20, x0:x0 + 24] = (torch.rand(20, 24, generator=g) > 0.5).float()
    x = torch.randn(B, C, H, W, generator=g) * occ
    w = torch.randn(32, C, 7, 7, generator=g) * 0.05
    bias = torch.randn(32, generator=g)
    spec = MC.ConvSpec(7, 7, 2, 3, False)
    xd = x.to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    dense, pd = MC.conv_forward(xd, w.cuda(), bias.cuda(), spec, out_relu=True, want_stats=True, out_dtype=torch.float32)
    sparse, ps = MC.conv_forward(xd, w.cuda(), bias.cuda(), spec, out_relu=True, want_stats=True, out_dtype=torch.float32,
                                 occupancy=occ.cuda().contiguous())
    assert torch.equal(dense, sparse) and torch.equal(pd[..., :32], ps[..., :32])  # (columns beyond the 32 channels are never written)
    assert float((dense - torch.relu(bias.cuda()).view(1, -1, 1, 1)).abs().amax()) > 0.1  # (not all-bias)


@pytest.mark.parametrize("geom", [(2, 64, 32, 128, 128, 7, 2, 3), (1, 16, 24, 45, 70, 3, 2, 1), (3, 64, 64, 33, 31, 7, 2, 3), (1, 8, 4, 64, 64, 5, 1, 2),
                                  (2, 64, 32, 512, 512, 7, 2, 3)])
@pytest.mark.parametrize("density", [0.02, 0.5, 0.0])
def test_sparse_canvas_weight_gradient_matches_fp64(geom, density):
    """liso_conv_wgrad_sparse_f32: the weight / bias gradient of a convolution on a sparse canvas (the encoders' 7x7 / 2 stem on the
    pillar canvas, extractor.py:230-232) from the occupied cells alone, against autograd in fp64 on the same canvas: the SLIM stem at
    the bench's size, ragged maps, other kernels / strides, a dense-ish and an empty canvas; bitwise reproducible."""
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W, k, s, p = geom
    g = torch.Generator().manual_seed(3)
    occ = (torch.rand(B, 1, H, W, generator=g) < density).float()
    x = torch.randn(B, Ci, H, W, generator=g) * occ
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    spec = MC.ConvSpec(k, k, s, p, False)
    x64, w64, b64 = x.double(), w.double().requires_grad_(True), b.double().requires_grad_(True)
    out = F.conv2d(x64, w64, b64, stride=s, padding=p)
    dy = torch.randn(out.shape, generator=g)
    gw, gb = torch.autograd.grad(out, [w64, b64], dy.double())
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    res = MC.conv_wgrad_sparse(xd, occ.cuda(), dyd, tuple(w.shape), spec)
    assert res is not None
    dw, db = res
    scale = max(float(gw.abs().max()), 1e-30)
    assert float((dw.double().cpu() - gw).abs().max()) <= 3e-6 * scale + 1e-12, float((dw.double().cpu() - gw).abs().max()) / scale
    assert _rel(db, gb) <= 1e-5
    dw2, db2 = MC.conv_wgrad_sparse(xd, occ.cuda(), dyd, tuple(w.shape), spec)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("relu", [False, True])
def test_stem_convolution_with_occupancy_trains_like_the_dense_path(relu):
    """mfma_conv.conv2d(layer, canvas, occupancy=...) in training: forward from the sparse stem kernel = the dense call up to the fp32
    summation order, weight / bias gradients from the sparse cell-list kernel within fp32 rounding of the dense F32X3 kernel's, input
    gradient from the sparse data-gradient kernel = the dense one's at the occupied cells"""
    from liso_amd.utils import mfma_conv as MC

    g = torch.Generator().manual_seed(5)
    occ = (torch.rand(2, 1, 128, 128, generator=g) < 0.03).float().cuda()
    x = (torch.randn(2, 64, 128, 128, generator=g).cuda() * occ).contiguous(memory_format=torch.channels_last)
    outs = []
    for use_occ in (False, True):
        torch.manual_seed(1)
        layer = torch.nn.Conv2d(64, 32, 7, stride=2, padding=3).cuda()
        xi = x.clone().requires_grad_(True)
        y = MC.conv2d(layer, xi, relu=relu, occupancy=occ if use_occ else None)
        (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        outs.append((y.detach(), layer.weight.grad.clone(), layer.bias.grad.clone(), xi.grad.clone()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) <= 2e-6 * float(outs[0][0].abs().max())
    # the input gradient comes from the sparse data-gradient kernel: the dense one's values at the occupied cells (other summation
    # order; with the ReLU a mask bit may flip where the output is within rounding of zero), exact zeros elsewhere
    m = occ.bool().expand_as(outs[0][3])
    dd, ds = outs[0][3], outs[1][3]
    assert float((ds - dd)[m].abs().max()) <= (1e-3 if relu else 2e-6) * float(dd.abs().max())
    assert float(ds[~m].abs().max()) == 0.0
    tol = 1e-3 if relu else 1e-4
    assert _rel(outs[1][1], outs[0][1]) <= tol and _rel(outs[1][2], outs[0][2]) <= tol


@pytest.mark.parametrize("B,ci_true,co,H,W,k", [(2, 4, 64, 64, 64, 7), (3, 2, 64, 17, 23, 7), (1, 3, 80, 9, 70, 7), (2, 4, 32, 16, 16, 5),
                                                 (12, 4, 64, 64, 64, 7)])
def test_small_channel_7x7_weight_gradient_matches_fp64(B, ci_true, co, H, W, k):
    """liso_conv_wgrad_smallci_f32 (the motion encoder's conv_flow1 / conv_class1, update.py:57,66: 7x7 on 2-4 channels) against
    torch's weight gradient in fp64: maps whose width is not a multiple of the window walk, output channel counts beside the
    64-lane tile, zero-padded input channels (their gradient rows must be exact zeros), bias gradient, bitwise repeatability"""
    from liso_amd.utils import mfma_conv as MC

    torch.manual_seed(B * 100 + W)
    x = torch.randn(B, 4, H, W, device="cuda")
    x[:, ci_true:] = 0
    x = x.contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, co, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    spec = MC.ConvSpec(k, k, 1, k // 2, False)
    dw, db = MC.conv_wgrad(x, dy, (co, 4, k, k), spec, want_bias=True)
    dw2, db2 = MC.conv_wgrad(x, dy, (co, 4, k, k), spec, want_bias=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    ref = torch.nn.grad.conv2d_weight(x.double(), (co, 4, k, k), dy.double(), stride=1, padding=k // 2)
    scale = float(ref.abs().max())
    assert float((dw.double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, (B * H * W) ** 0.5 / 30)
    assert float(dw[:, ci_true:].abs().max() if ci_true < 4 else 0.0) == 0.0
    refb = dy.double().sum(dim=(0, 2, 3))
    assert float((db.double() - refb).abs().max()) <= 1e-5 * float(refb.abs().max())


@pytest.mark.parametrize("B,H,W,density", [(2, 128, 128, 0.02), (1, 64, 192, 0.3), (3, 512, 512, 0.016), (2, 64, 64, 0.0), (1, 32, 64, 1.0)])
@pytest.mark.parametrize("norm_kind", ["instance", "none"])
def test_sparse_stem_convolution_equals_the_dense_kernel(B, H, W, density, norm_kind, monkeypatch):
    """liso_sparse_stem_forward_f32 (the encoders' 7x7 / 2 stem on the pillar canvas, only occupied cells multiplied) against the dense
    F32X3 kernel on the same canvas: same raw output up to the fp32 summation order (the dense kernel adds all taps into one
    accumulator, the sparse one adds per-tap products), pixels whose window is empty equal the bias bit for bit, the pending
    InstanceNorm statistics agree, border cells / odd columns / full and empty canvases included; the overflow flag stays clear"""
    from liso_amd.utils import mfma_conv as MC

    prev = MC.set_fp32_mode("x3")
    try:
        torch.manual_seed(H + W)
        occ = (torch.rand(B, 1, H, W, device="cuda") < density).float()
        occ[:, :, 0, 0] = 1.0 if density > 0 else 0.0
        occ[:, :, H - 1, W - 1] = 1.0 if density > 0 else 0.0
        x = (torch.randn(B, 64, H, W, device="cuda") * occ).contiguous(memory_format=torch.channels_last)
        conv = torch.nn.Conv2d(64, 32, 7, stride=2, padding=3).cuda()
        norm = torch.nn.InstanceNorm2d(32, eps=1e-3, affine=True).cuda() if norm_kind == "instance" else torch.nn.Sequential()
        for p in conv.parameters():
            p.requires_grad_(False)
        monkeypatch.setenv("LISO_SPARSE_STEM", "0")
        yd, fd = MC.conv_in(x, None, conv, norm, occupancy=occ)
        monkeypatch.setenv("LISO_SPARSE_STEM", "1")
        ys, fs = MC.conv_in(x, None, conv, norm, occupancy=occ)
        ys2, _ = MC.conv_in(x, None, conv, norm, occupancy=occ)
    finally:
        MC.set_fp32_mode(prev)
    assert torch.equal(ys, ys2)
    assert ys.shape == yd.shape == (B, 32, H // 2, W // 2)
    scale = float(yd.abs().max())
    assert float((ys - yd).abs().max()) <= 2e-6 * max(scale, 1.0)
    win = torch.nn.functional.max_pool2d(occ, 7, stride=2, padding=3)  # 1 where the output pixel's window holds a cell
    raw = conv.bias.view(1, 32, 1, 1).expand_as(ys)
    if norm_kind == "none":
        raw = torch.relu(raw)
    empty = (win == 0).expand_as(ys)
    assert torch.equal(ys[empty], raw[empty])
    if norm_kind == "instance":
        assert float((fs.stats - fd.stats).abs().max()) <= 1e-4 * float(fd.stats.abs().max())
    assert not MC.sparse_stem_overflowed(x.device)


def test_sparse_stem_reports_a_batch_beyond_its_cell_capacity(monkeypatch):
    from liso_amd.utils import mfma_conv as MC

    prev = MC.set_fp32_mode("x3")
    try:
        occ = (torch.rand(1, 1, 64, 64, device="cuda") < 0.5).float()
        x = (torch.randn(1, 64, 64, 64, device="cuda") * occ).contiguous(memory_format=torch.channels_last)
        conv = torch.nn.Conv2d(64, 32, 7, stride=2, padding=3).cuda().requires_grad_(False)
        monkeypatch.setattr(MC, "SPARSE_STEM_MAX_CELLS", 64)
        MC.conv_in(x, None, conv, torch.nn.Sequential(), occupancy=occ)
        assert MC.sparse_stem_overflowed(x.device)
        MC._SPARSE_OVERFLOW[x.device.index].zero_()
    finally:
        MC.set_fp32_mode(prev)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,W,density", [(2, 128, 128, 0.03), (1, 64, 192, 0.4), (2, 512, 512, 0.016), (1, 64, 64, 0.0)])
def test_sparse_first_rpn_layer_equals_the_dense_kernels(B, H, W, density, dtype, monkeypatch):
    """3x3 / stride 2 / padding 1, 64 -> 64 channels on the pillar canvas with its occupancy map (the detector's first layer,
    rpn.py:113-131) through fused_conv with BatchNorm statistics: raw output, batch statistics and the data gradient AT THE OCCUPIED
    CELLS from the sparse kernels against the dense kernels on the same tensors (bf16 tensors and fp32 tensors in F32X3 arithmetic)"""
    from liso_amd.utils import mfma_conv as MC

    prev = MC.set_fp32_mode("x3")
    try:
        torch.manual_seed(H + W)
        occ = (torch.rand(B, 1, H, W, device="cuda") < density).float()
        if density > 0:
            occ[:, :, 0, 0] = 1.0
            occ[:, :, H - 1, W - 1] = 1.0
            occ[:, :, 0, W - 1] = 1.0
        x0 = (torch.randn(B, 64, H, W, device="cuda") * occ).to(dtype).contiguous(memory_format=torch.channels_last)
        conv = torch.nn.Conv2d(64, 64, 3, stride=2, bias=False).cuda()
        bn = torch.nn.BatchNorm2d(64, eps=1e-3, momentum=0.01).cuda().train()
        spec = MC.ConvSpec(3, 3, 2, 1)
        g0 = torch.randn(B, 64, H // 2, W // 2, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        res = []
        for sparse in ("0", "1"):
            monkeypatch.setenv("LISO_SPARSE_STEM", sparse)
            bn.running_mean.zero_(), bn.running_var.fill_(1.0)
            x = x0.clone().requires_grad_(True)
            conv.weight.grad = None
            y, fold = MC.fused_conv(x, None, conv, out_bn=bn, spec=spec, occupancy=occ)
            (y.float() * g0.float()).sum().backward()
            res.append((y.detach().float(), fold.groups[0]["stats"].clone(), x.grad.float().clone(), conv.weight.grad.clone()))
    finally:
        MC.set_fp32_mode(prev)
    (yd, sd, gd, wd), (ys, ss, gs, ws) = res
    tol = 1e-2 if dtype == torch.bfloat16 else 2e-6  # (bf16: a sum that lands on a rounding boundary may round the other way)
    assert float((ys - yd).abs().max()) <= tol * max(float(yd.abs().max()), 1.0)
    assert float((ss - sd).abs().max()) <= 1e-3 * max(float(sd.abs().max()), 1e-6)
    m = occ.bool().expand_as(gd)
    assert float((gs - gd)[m].abs().max() if density > 0 else 0.0) <= tol * max(float(gd.abs().max()), 1.0)
    assert float(gs[~m].abs().max() if (~m).any() else 0.0) == 0.0
    assert _rel(ws, wd) <= 1e-2 if dtype == torch.bfloat16 else _rel(ws, wd) <= 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_convolutions_writing_channel_ranges_of_one_buffer_and_slice_cat(dtype):
    """fused_conv(..., out=(buffer, first_channel)): three convolutions (3x3, 1x1, transposed 2x2 -- the deblock geometries of
    rpn.py:52-63) write one concatenated map without a concatenation pass; `slice_cat` connects it to autograd: same values as
    torch.cat of the stand-alone outputs, same input / weight gradients"""
    from liso_amd.utils import mfma_conv as MC

    torch.manual_seed(9)
    B = 2
    xs = [torch.randn(B, 64, 32, 32, device="cuda"), torch.randn(B, 128, 32, 32, device="cuda"), torch.randn(B, 64, 16, 16, device="cuda")]
    xs = [t.to(dtype).contiguous(memory_format=torch.channels_last) for t in xs]
    convs = [torch.nn.Conv2d(64, 128, 3, padding=1, bias=False).cuda(), torch.nn.Conv2d(128, 64, 1, bias=False).cuda(),
             torch.nn.ConvTranspose2d(64, 128, 2, stride=2, bias=False).cuda()]
    chans = [128, 64, 128]
    w = torch.randn(B, sum(chans), 32, 32, device="cuda")
    res = []
    for in_place in (False, True):
        ins = [t.clone().requires_grad_(True) for t in xs]
        for c in convs:
            c.weight.grad = None
        if in_place:
            buf = torch.full((B, 32, 32, sum(chans)), float("nan"), dtype=dtype, device="cuda")
            parts, off = [], 0
            for t, c, n in zip(ins, convs, chans):
                parts.append(MC.fused_conv(t, None, c, out=(buf, off))[0])
                off += n
            y = MC.slice_cat(buf.permute(0, 3, 1, 2), parts)
        else:
            y = torch.cat([MC.fused_conv(t, None, c)[0] for t, c in zip(ins, convs)], dim=1)
        (y.float() * w).sum().backward()
        res.append((y.detach().float(), [t.grad.float() for t in ins], [c.weight.grad.clone() for c in convs]))
    (y0, gx0, gw0), (y1, gx1, gw1) = res
    assert torch.equal(y0, y1)
    for a, b in zip(gx0, gx1):
        assert torch.equal(a, b)
    for a, b in zip(gw0, gw1):
        assert torch.equal(a, b)
