"""GPU parity of conv_roles_kernel (the 3x3 / stride-1 kernel with loader waves + MFMA waves, persistent blocks) against torch's
convolution in fp64 on the same (rounded) operands, on the shapes that stress ITS machinery rather than the networks' layers: more tiles
than CUs (several tiles per persistent block), tile counts that are no multiple of 8, partial tiles on every edge, channel counts
that are no multiple of the slab (16 fp32 / 32 bf16) or of the 32 / 64 / 96-channel panels, the prologue + statistics path, channel-slice
outputs, the data gradient (mirrored taps), forced tile shapes.  Every case first checks that the descriptor really takes this kernel."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (B, Ci, Co, H, W)
SHAPES = [
    (1, 32, 32, 5, 7),        # one partial tile, two fp32 slabs (the kernel needs >= 2 slabs per tile: bf16 skips this one)
    (2, 32, 32, 37, 45),      # partial tiles on both edges
    (3, 48, 100, 33, 65),     # 100 filters: 96 + 4 / 64 + 36, three slabs of 16
    (1, 20, 7, 64, 64),       # Ci not a multiple of the slab (fp32 only), 7 filters: narrow-store epilogue
    (5, 64, 64, 96, 96),      # 5 * 24 * 3 = 360 tiles at 4 rows: more than 256 persistent blocks
    (2, 96, 192, 64, 64),     # 96-channel panels
    (7, 32, 96, 31, 33),      # odd tile counts
    (1, 304, 96, 16, 16),     # deep reduction, few tiles (gridDim < 8)
]
ARITH = [pytest.param(torch.float32, id="f32x3"), pytest.param(torch.bfloat16, id="bf16")]


def _mk(shape, dtype, seed=0):
    B, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    return x, w, b


def _kind(x, w, spec):
    from liso_amd import _lib as L
    from liso_amd.utils import mfma_conv as MC

    xv, xps = MC.as_nhwc(x, MC._vec(MC._mode(x.dtype)))
    B, hi, wi, ci = xv.shape
    co = w.shape[0]
    d = MC.gather_desc(spec, B, hi, wi, ci, xps, hi, wi, co, co, 0, MC._mode(x.dtype), x.dtype == torch.float32, False, False)
    return L.lib().liso_conv_kernel_kind(ctypes.byref(d))


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-12))


@pytest.mark.parametrize("dtype", ARITH)
@pytest.mark.parametrize("shape", SHAPES)
def test_forward_prologue_statistics_and_slices(shape, dtype):
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W = shape
    if dtype == torch.bfloat16 and (Ci % 32 or Ci < 64 or Co % 2):
        pytest.skip("bf16 layers take this kernel only with whole 32-channel slabs, at least two of them")
    x, w, b = _mk(shape, dtype)
    spec = MC.ConvSpec(3, 3, 1, 1, False)
    xd = x.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    wd, bd = w.to(DEV), b.to(DEV)
    assert _kind(xd, wd, spec) == 1, "descriptor does not take conv_roles_kernel"
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5  # bf16: the OUTPUT is rounded to bf16 (2^-9), fp32: 2^-16 per product
    # plain, with bias + ReLU
    y, _ = MC.conv_forward(xd, wd, bd, spec, out_relu=True)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    assert _rel(y.float(), ref) <= tol
    # prologue (per-channel affine + ReLU of the producer) and the statistics epilogue
    g = torch.Generator().manual_seed(1)
    sc, sh = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    y2, part = MC.conv_forward(xd, wd, None, spec, sc.to(DEV), sh.to(DEV), in_relu=True, want_stats=True)
    xin = F.relu(x.double() * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    if dtype == torch.bfloat16:
        xin = xin.float().bfloat16().double()
    ref2 = F.conv2d(xin, w.double(), None, padding=1)
    assert _rel(y2.float(), ref2) <= tol
    stored = y2.float().double()  # the sums are taken over the STORED (rounded) values
    s1 = part[:, 0, :Co].double().sum(0).cpu()
    s2 = part[:, 1, :Co].double().sum(0).cpu()
    assert torch.allclose(s1, stored.sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-3 * float(stored.abs().max()))
    assert torch.allclose(s2, stored.square().sum((0, 2, 3)).cpu(), rtol=1e-4, atol=1e-6)
    assert torch.isfinite(part).all()  # every statistics row is written, the ones of partial tiles included
    # channel-slice output: the result becomes channels [8, 8 + Co) of a wider buffer, its neighbours stay untouched
    if Co % 8 == 0:
        buf = torch.full((B, H, W, Co + 24), 7.0, dtype=dtype, device=DEV)
        y3, _ = MC.conv_forward(xd, wd, bd, spec, out=(buf, 8))
        assert _rel(y3.float(), F.conv2d(x.double(), w.double(), b.double(), padding=1)) <= tol
        assert bool((buf[..., :8] == 7.0).all()) and bool((buf[..., 8 + Co:] == 7.0).all())


@pytest.mark.parametrize("dtype", ARITH)
@pytest.mark.parametrize("shape", [(2, 64, 96, 40, 72), (1, 32, 64, 9, 33), (3, 128, 32, 64, 64)])
def test_data_gradient(shape, dtype):
    from liso_amd.utils import mfma_conv as MC

    B, Ci, Co, H, W = shape
    x, w, _ = _mk(shape, dtype)
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(B, Co, H, W, generator=g)
    if dtype == torch.bfloat16:
        dy = dy.bfloat16().float()
    spec = MC.ConvSpec(3, 3, 1, 1, False)
    dyd = dy.to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    gx = MC.conv_dgrad(dyd, w.to(DEV), spec, (B, Ci, H, W))
    x64 = x.double().requires_grad_(True)
    (ref,) = torch.autograd.grad(F.conv2d(x64, w.double(), padding=1), [x64], dy.double())
    assert _rel(gx.float(), ref) <= (2e-2 if dtype == torch.bfloat16 else 2e-5)


def test_forced_tile_shapes_agree_bitwise_on_statistics_and_closely_on_values():
    """LISO_ROLES_MI / LISO_ROLES_NJ force the tile shape (a plan-time switch: one child process per shape).  Values must agree to
    fp32 rounding of the summation order (identical here: the order over (slab, tap) does not depend on the tile), and the statistics
    rows -- fixed 4 x 32-pixel sets, fixed tree -- bit for bit."""
    code = r"""
import sys, torch
sys.path.insert(0, %r)
from liso_amd.utils import mfma_conv as MC
torch.manual_seed(0)
x = torch.randn(3, 48, 40, 72, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 48, 3, 3, device="cuda") * 0.05
y, part = MC.conv_forward(x, w, None, MC.ConvSpec(3, 3, 1, 1, False), want_stats=True)
torch.save({"y": y.cpu(), "s": part[:, :, :64].cpu(), "rows": part.shape[0]}, sys.argv[1])
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for mi, nj in ((1, 1), (1, 2), (2, 1), (2, 2), (1, 3)):
        path = f"/tmp/liso_roles_forced_{mi}{nj}.pt"
        env = dict(os.environ, LISO_ROLES_MI=str(mi), LISO_ROLES_NJ=str(nj))
        r = subprocess.run([sys.executable, "-c", code % root, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(torch.load(path))
    for k, o in enumerate(outs[1:]):
        print("shape", k + 1)
        assert torch.equal(o["y"], outs[0]["y"])
        assert o["rows"] == outs[0]["rows"]
        bad = (o["s"] != outs[0]["s"]).any(dim=2).any(dim=1).nonzero().flatten().tolist()
        if bad:
            r0 = bad[0]
            diff = (o["s"][r0] - outs[0]["s"][r0])
            ch = diff.abs().sum(0).nonzero().flatten().tolist()
            raise AssertionError((len(bad), bad[:10], "channels", ch[:12], o["s"][r0][:, ch[:4]].tolist(), outs[0]["s"][r0][:, ch[:4]].tolist()))
