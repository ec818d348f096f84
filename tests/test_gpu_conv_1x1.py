"""GPU parity of conv_1x1_kernel (1x1 convolutions of stride 1 / 2 on fp32 tensors in the F32X3 arithmetic: fragments straight from global
memory, no LDS staging) against torch's convolution in fp64 -- the layers of the reference that take it (liso/slim/model/extractor.py's
residual shortcuts, update.py:49 conv_stat_corr1 with its 196 = 12 x 16 + 4 channels) and the shapes that stress its edges: channel counts
that are no multiple of the 16-channel step or of the 32 / 64 / 96-filter panels, partial tiles, the producer's pending per-channel and
per-sample affine + ReLU, the statistics epilogue, channel-slice outputs, the data gradient."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (B, Ci, Co, H, W, stride)
SHAPES = [
    (4, 196, 96, 64, 64, 1),     # conv_stat_corr1 at 4 sweep pairs
    (8, 32, 64, 256, 256, 2),    # encoder shortcut, stride 2, 8-row tiles
    (8, 64, 96, 128, 128, 2),
    (2, 96, 160, 64, 64, 1),     # 160 filters: panels of 96 + 64
    (1, 4, 7, 5, 9, 1),          # one partial tile, 4 channels, 7 filters
    (3, 20, 33, 33, 65, 1),      # nothing is a multiple of anything
    (2, 100, 128, 31, 47, 2),    # odd map, stride 2
    (1, 400, 32, 16, 16, 1),     # deep reduction, few tiles
]


def _mk(shape, seed=0):
    B, Ci, Co, H, W, _ = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    return x, w, b


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


def _kind(xd, Co, spec):
    from liso_amd import _lib as L
    from liso_amd.utils import mfma_conv as MC

    xv, xps = MC.as_nhwc(xd, 4)
    B, hi, wi, ci = xv.shape
    ho, wo = spec.out_hw(hi, wi)
    d = MC.gather_desc(spec, B, hi, wi, ci, xps, ho, wo, Co, Co, 0, L.CONV_F32X3, True, False, False)
    return L.lib().liso_conv_kernel_kind(ctypes.byref(d))


@pytest.mark.parametrize("shape", SHAPES)
def test_forward_prologue_statistics_and_slices(shape):
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W, st = shape
    x, w, b = _mk(shape)
    spec = MC.ConvSpec(1, 1, st, 0, False)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    wd, bd = w.to(DEV), b.to(DEV)
    assert _kind(xd, Co, spec) == 2, "descriptor does not take conv_1x1_kernel"
    tol = 2e-5
    y, _ = MC.conv_forward(xd, wd, bd, spec, out_relu=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=st))
    assert y.shape == ref.shape and _rel(y, ref) <= tol
    # the producer's pending affine + ReLU (one vector for the batch: BatchNorm) and the statistics epilogue
    g = torch.Generator().manual_seed(1)
    sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    y2, part = MC.conv_forward(xd, wd, None, spec, sc.to(DEV), sh.to(DEV), in_relu=True, want_stats=True)
    xin = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    ref2 = F.conv2d(xin, w.double(), None, stride=st)
    assert _rel(y2, ref2) <= tol
    stored = y2.double()
    s1, s2 = part[:, 0, :Co].double().sum(0).cpu(), part[:, 1, :Co].double().sum(0).cpu()
    assert torch.allclose(s1, stored.sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-3 * float(stored.abs().max()))
    assert torch.allclose(s2, stored.square().sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-6)
    assert torch.isfinite(part).all()
    # one vector per sample (InstanceNorm), no ReLU
    scb, shb = torch.rand(B, Ci, generator=g) + 0.5, torch.randn(B, Ci, generator=g) * 0.2
    y3, _ = MC.conv_forward(xd, wd, bd, spec, scb.to(DEV).contiguous(), shb.to(DEV).contiguous(), in_relu=False, affine_batch_stride=Ci)
    xin3 = x.double() * scb.double()[:, :, None, None] + shb.double()[:, :, None, None]
    assert _rel(y3, F.conv2d(xin3, w.double(), b.double(), stride=st)) <= tol
    # channel-slice output
    if Co % 8 == 0:
        ho, wo = y.shape[2], y.shape[3]
        buf = torch.full((B, ho, wo, Co + 24), 7.0, dtype=torch.float32, device=DEV)
        y4, _ = MC.conv_forward(xd, wd, bd, spec, out=(buf, 8))
        assert _rel(y4, F.conv2d(x.double(), w.double(), b.double(), stride=st)) <= tol
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + Co:] == 7.0).all())
    # an input that is a channel slice of a wider tensor (pixel stride > channels)
    if Ci % 4 == 0:
        wide = torch.randn(B, H, W, Ci + 12, generator=g).to(DEV)
        xs = wide[..., 4:4 + Ci].permute(0, 3, 1, 2)
        y5, _ = MC.conv_forward(xs, wd, None, spec)
        assert _rel(y5, F.conv2d(xs.double().cpu(), w.double(), None, stride=st)) <= tol


@pytest.mark.parametrize("shape", [(2, 96, 64, 40, 72, 1), (1, 196, 96, 16, 16, 1), (3, 32, 128, 33, 31, 1)])
def test_data_gradient_of_stride_1_layers(shape):
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W, st = shape
    x, w, _ = _mk(shape)
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(B, Co, H, W, generator=g)
    spec = MC.ConvSpec(1, 1, 1, 0, False)
    gx = MC.conv_dgrad(dy.to(DEV).contiguous(memory_format=torch.channels_last), w.to(DEV), spec, (B, Ci, H, W))
    x64 = x.double().requires_grad_(True)
    (ref,) = torch.autograd.grad(F.conv2d(x64, w.double()), [x64], dy.double())
    assert _rel(gx, ref) <= 2e-5


def test_run_to_run_bitwise_and_not_a_number_free_padding():
    """two launches agree bit for bit; rows of a partial tile and channels beyond ci never reach the result (the tensor's first bytes are
    read in their place: poisoned here)"""
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W = 2, 20, 40, 9, 37
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Ci, H, W, generator=g)
    x[0, :, 0, 0] = float("nan")  # the pixel whose address stands in for every masked load
    w = torch.randn(Co, Ci, 1, 1, generator=g)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    spec = MC.ConvSpec(1, 1, 1, 0, False)
    y1, _ = MC.conv_forward(xd, w.to(DEV), None, spec)
    y2, _ = MC.conv_forward(xd, w.to(DEV), None, spec)
    assert torch.equal(y1.cpu().nan_to_num(1234.0), y2.cpu().nan_to_num(1234.0))
    bad = torch.isnan(y1).any(dim=1)
    assert int(bad.sum()) == 1 and bool(bad[0, 0, 0])  # only the poisoned pixel's own outputs


# ---- conv_taps_kernel: windows on 2-8 input channels (two taps per MFMA step) -----------------------------------------------------------
# (B, Ci, Co, H, W, k, stride)
TAP_SHAPES = [
    (4, 8, 128, 64, 64, 7, 1),    # the motion encoder's merged conv_class1 | conv_flow1 on the packed state pixel (update.py:54-59)
    (2, 4, 64, 33, 45, 7, 1),     # conv_class1 alone, partial tiles
    (1, 4, 32, 16, 16, 5, 1),
    (3, 8, 100, 20, 70, 3, 1),    # 9 taps: the odd count's last half step
    (2, 8, 40, 31, 47, 3, 2),     # stride 2
    (1, 4, 7, 5, 9, 7, 1),        # one partial tile, windows larger than the map
]


@pytest.mark.parametrize("shape", TAP_SHAPES)
def test_small_channel_windows(shape):
    from liso_amd import _lib as L
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W, k, st = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    spec = MC.ConvSpec(k, k, st, k // 2, False)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last)
    xv, xps = MC.as_nhwc(xd, 4)
    ho, wo = spec.out_hw(H, W)
    d = MC.gather_desc(spec, B, H, W, Ci, xps, ho, wo, Co, Co, 0, L.CONV_F32X3, True, False, False)
    assert L.lib().liso_conv_kernel_kind(ctypes.byref(d)) == 3, "descriptor does not take conv_taps_kernel"
    y, _ = MC.conv_forward(xd, w.to(DEV), b.to(DEV), spec, out_relu=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=k // 2))
    assert y.shape == ref.shape and _rel(y, ref) <= 2e-5
    sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    y2, part = MC.conv_forward(xd, w.to(DEV), None, spec, sc.to(DEV), sh.to(DEV), in_relu=True, want_stats=True)
    xin = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    assert _rel(y2, F.conv2d(xin, w.double(), None, stride=st, padding=k // 2)) <= 2e-5
    stored = y2.double()
    s1, s2 = part[:, 0, :Co].double().sum(0).cpu(), part[:, 1, :Co].double().sum(0).cpu()
    assert torch.allclose(s1, stored.sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-3 * float(stored.abs().max()))
    assert torch.allclose(s2, stored.square().sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-6)
    if Co % 8 == 0:
        buf = torch.full((B, ho, wo, Co + 16), 7.0, dtype=torch.float32, device=DEV)
        y3, _ = MC.conv_forward(xd, w.to(DEV), b.to(DEV), spec, out=(buf, 8))
        assert _rel(y3, F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=k // 2)) <= 2e-5
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + Co:] == 7.0).all())
    y4, _ = MC.conv_forward(xd, w.to(DEV), b.to(DEV), spec, out_relu=True)
    assert torch.equal(y, y4)


# ---- the same kernel on bf16 tensors and on tap classes (transposed convolutions with kernel = stride: rpn.py:70-104 deblocks) --------------
@pytest.mark.parametrize("dtype", [pytest.param(torch.float32, id="f32x3"), pytest.param(torch.bfloat16, id="bf16")])
@pytest.mark.parametrize("shape", [(2, 128, 128, 32, 32, 2), (2, 256, 128, 64, 64, 2), (1, 64, 96, 9, 37, 2), (3, 40, 72, 20, 33, 1), (2, 128, 128, 128, 128, 1)])
def test_transposed_kernel_equals_stride_and_bf16_1x1(shape, dtype):
    from liso_amd import _lib as L
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W, k = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, H, W, generator=g)
    transposed = k > 1
    w = (torch.randn(Ci, Co, k, k, generator=g) if transposed else torch.randn(Co, Ci, 1, 1, generator=g)) / Ci ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    spec = MC.ConvSpec(k, k, k, 0, transposed)
    xd = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    mode = MC._mode(dtype)
    xv, xps = MC.as_nhwc(xd, MC._vec(mode))
    ho, wo = spec.out_hw(H, W)
    build = MC.scatter_desc if transposed else MC.gather_desc
    d = build(spec, B, H, W, Ci, xps, ho, wo, Co, Co, 0, mode, dtype == torch.float32, False, False)
    assert L.lib().liso_conv_kernel_kind(ctypes.byref(d)) == 2, "descriptor does not take conv_1x1_kernel"
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    ref_fn = (lambda xx, bb: F.conv_transpose2d(xx, w.double(), bb, stride=k)) if transposed else (lambda xx, bb: F.conv2d(xx, w.double(), bb))
    y, _ = MC.conv_forward(xd, w.to(DEV), b.to(DEV), spec, out_relu=True)
    ref = F.relu(ref_fn(x.double(), b.double()))
    assert y.shape == ref.shape and _rel(y.float(), ref) <= tol
    sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    y2, part = MC.conv_forward(xd, w.to(DEV), None, spec, sc.to(DEV), sh.to(DEV), in_relu=True, want_stats=True)
    xin = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    if dtype == torch.bfloat16:
        xin = xin.float().bfloat16().double()
    assert _rel(y2.float(), ref_fn(xin, None)) <= tol
    stored = y2.float().double()
    s1, s2 = part[:, 0, :Co].double().sum(0).cpu(), part[:, 1, :Co].double().sum(0).cpu()
    assert torch.allclose(s1, stored.sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-3 * float(stored.abs().max()))
    assert torch.allclose(s2, stored.square().sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-6)
    assert torch.isfinite(part).all()


@pytest.mark.parametrize("dtype", [pytest.param(torch.float32, id="f32x3"), pytest.param(torch.bfloat16, id="bf16")])
@pytest.mark.parametrize("shape", [(2, 64, 128, 256, 256, 2), (1, 32, 40, 18, 70, 2), (2, 16, 32, 27, 27, 3)])
def test_kernel_equals_stride_convolutions(shape, dtype):
    """k x k / stride k (the k = 2 deblock): non-overlapping windows -- one 1x1 problem on k * k * Ci channels, no pixel read twice"""
    from liso_amd import _lib as L
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W, k = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    spec = MC.ConvSpec(k, k, k, 0, False)
    xd = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    mode = MC._mode(dtype)
    xv, xps = MC.as_nhwc(xd, MC._vec(mode))
    ho, wo = spec.out_hw(H, W)
    d = MC.gather_desc(spec, B, H, W, Ci, xps, ho, wo, Co, Co, 0, mode, dtype == torch.float32, False, False)
    assert L.lib().liso_conv_kernel_kind(ctypes.byref(d)) == 2, "descriptor does not take conv_1x1_kernel"
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    y, _ = MC.conv_forward(xd, w.to(DEV), b.to(DEV), spec, out_relu=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=k))
    assert y.shape == ref.shape and _rel(y.float(), ref) <= tol
    sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    y2, part = MC.conv_forward(xd, w.to(DEV), None, spec, sc.to(DEV), sh.to(DEV), in_relu=True, want_stats=True)
    xin = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    if dtype == torch.bfloat16:
        xin = xin.float().bfloat16().double()
    assert _rel(y2.float(), F.conv2d(xin, w.double(), None, stride=k)) <= tol
    stored = y2.float().double()
    assert torch.allclose(part[:, 0, :Co].double().sum(0).cpu(), stored.sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-3 * float(stored.abs().max()))


@pytest.mark.parametrize("dtype", [pytest.param(torch.float32, id="f32x3"), pytest.param(torch.bfloat16, id="bf16")])
@pytest.mark.parametrize("shape", [(8, 32, 64, 256, 256), (4, 64, 96, 128, 128), (2, 64, 128, 37, 45), (1, 16, 24, 9, 70)])
def test_3x3_stride_2_layers_on_small_pixels(shape, dtype):
    """3x3 / stride 2 / padding 1 on pixels of <= 256 bytes (the encoders' downsampling layers, extractor.py:211-297; the detector's block
    entries, rpn.py:113-131): the direct kernel re-reads each pixel 2.25 times through L1 instead of staging tiles"""
    from liso_amd import _lib as L
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    spec = MC.ConvSpec(3, 3, 2, 1, False)
    xd = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    mode = MC._mode(dtype)
    xv, xps = MC.as_nhwc(xd, MC._vec(mode))
    ho, wo = spec.out_hw(H, W)
    d = MC.gather_desc(spec, B, H, W, Ci, xps, ho, wo, Co, Co, 0, mode, dtype == torch.float32, False, False)
    assert L.lib().liso_conv_kernel_kind(ctypes.byref(d)) == 2, "descriptor does not take conv_1x1_kernel"
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    y, _ = MC.conv_forward(xd, w.to(DEV), b.to(DEV), spec, out_relu=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    assert y.shape == ref.shape and _rel(y.float(), ref) <= tol
    sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    y2, part = MC.conv_forward(xd, w.to(DEV), None, spec, sc.to(DEV), sh.to(DEV), in_relu=True, want_stats=True)
    xin = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    if dtype == torch.bfloat16:
        xin = xin.float().bfloat16().double()
    assert _rel(y2.float(), F.conv2d(xin, w.double(), None, stride=2, padding=1)) <= tol
    stored = y2.float().double()
    assert torch.allclose(part[:, 0, :Co].double().sum(0).cpu(), stored.sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-3 * float(stored.abs().max()))
    assert torch.allclose(part[:, 1, :Co].double().sum(0).cpu(), stored.square().sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-6)
