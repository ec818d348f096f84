"""GPU parity: the gfx950 IoU / NMS kernels (through the C ABI) vs the CPU oracle and the committed goldens."""
import glob
import os

import numpy as np
import pytest

from oracle import iou3d as O

pytestmark = pytest.mark.gpu

TOL = 1e-5  # abs; device sin/cos/atan2 differ from glibc by <= 1 ulp, everything else is bit-identical


@pytest.fixture(scope="module")
def M():
    import torch

    assert torch.cuda.is_available()
    from liso_amd import iou3d_nms_cuda

    return iou3d_nms_cuda


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _nms(M, b, t, normal=False):
    import torch

    keep = torch.zeros(len(b), dtype=torch.int64)
    n = (M.nms_normal_gpu if normal else M.nms_gpu)(_dev(b), keep, t)
    return keep[:n].numpy()


def test_goldens(M, golden_dir):
    import torch

    for f in sorted(glob.glob(os.path.join(golden_dir, "iou3d_*.npz"))):
        g = np.load(f)
        b = g["boxes_sorted"]
        tb = _dev(b)
        n = len(b)
        iou = torch.zeros(n, n, device="cuda")
        ov = torch.zeros(n, n, device="cuda")
        assert M.boxes_iou_bev_gpu(tb, tb, iou) == 1
        assert M.boxes_overlap_bev_gpu(tb, tb, ov) == 1
        iou, ov = iou.cpu().numpy(), ov.cpu().numpy()
        assert np.abs(iou - g["iou"]).max() < TOL, f
        assert np.abs(ov - g["overlap"]).max() < 1e-4, f
        # exact zeros must be exact zeros (early-out is exact)
        assert np.array_equal(iou == 0, g["iou"] == 0), f
        frac_bitexact = (iou.view(np.uint32) == g["iou"].view(np.uint32)).mean()
        assert frac_bitexact > 0.99, (f, frac_bitexact)  # glibc sinf/cosf are not correctly rounded (and CPU-dependent)
        for t in (0.1, 0.3, 0.7):
            assert np.array_equal(_nms(M, b, t), g[f"keep_{int(t*100):03d}"]), (f, t)


@pytest.mark.parametrize("n,m,spread,seed", [(1, 1, 1.0, 0), (63, 65, 5.0, 1), (64, 64, 5.0, 2), (257, 130, 20.0, 3),
                                             (1000, 1000, 50.0, 4), (17, 2049, 30.0, 5)])
def test_iou_matrix_vs_oracle(M, n, m, spread, seed):
    import torch

    a, _ = O.random_boxes(n, seed, spread)
    b, _ = O.random_boxes(m, seed + 50, spread)
    out = torch.zeros(n, m, device="cuda")
    M.boxes_iou_bev_gpu(_dev(a), _dev(b), out)
    ref = O.boxes_iou_bev(a, b)
    assert np.abs(out.cpu().numpy() - ref).max() < TOL
    M.boxes_overlap_bev_gpu(_dev(a), _dev(b), out)
    assert np.abs(out.cpu().numpy() - O.boxes_overlap_bev(a, b)).max() < 1e-4


@pytest.mark.parametrize("n,spread,seed", [(1, 1.0, 0), (2, 1.0, 1), (64, 4.0, 2), (65, 4.0, 3), (128, 6.0, 4),
                                           (1000, 20.0, 5), (1000, 50.0, 6), (4096, 60.0, 7), (7000, 120.0, 8)])
def test_nms_selection_identical(M, n, spread, seed):
    b, s = O.random_boxes(n, seed, spread)
    b = b[np.argsort(-s, kind="stable")]
    for t in (0.1, 0.5):
        assert np.array_equal(_nms(M, b, t), O.nms(b, t)), (n, t)
    assert np.array_equal(_nms(M, b, 0.1, normal=True), O.nms_normal(b, 0.1))


def test_empty_and_device_resident(M):
    import torch

    e = torch.zeros(0, 7, device="cuda")
    keep = torch.zeros(0, dtype=torch.int64)
    assert M.nms_gpu(e, keep, 0.1) == 0
    out = torch.zeros(0, 5, device="cuda")
    assert M.boxes_iou_bev_gpu(e, torch.zeros(5, 7, device="cuda"), out) == 1
    b, s = O.random_boxes(500, 12, 10.0)
    b = b[np.argsort(-s, kind="stable")]
    kd, nd = M.nms_gpu_device(_dev(b), 0.1)
    k = kd[: int(nd.item())].cpu().numpy()
    assert np.array_equal(k, O.nms(b, 0.1))


def test_full_size_properties(M):
    """BASELINE-size properties that need no oracle: symmetry-free invariants of NMS at N=4096 dense."""
    import torch

    b, s = O.random_boxes(4096, 99, 40.0)
    b = b[np.argsort(-s, kind="stable")]
    tb = _dev(b)
    iou = torch.zeros(4096, 4096, device="cuda")
    M.boxes_iou_bev_gpu(tb, tb, iou)
    iou = iou.cpu().numpy()
    assert np.abs(np.diag(iou) - 1).max() < 1e-4
    assert np.abs(iou - iou.T).max() < 1e-4  # IoU is symmetric up to vertex-order rounding
    k = _nms(M, b, 0.1)
    assert np.array_equal(k, O.nms_from_iou(iou, 0.1)) or len(np.setxor1d(k, O.nms_from_iou(iou, 0.1))) <= 2
    sub = np.triu(iou[np.ix_(k, k)], 1)
    assert (sub <= 0.1 + 1e-5).all()
    # idempotence: NMS of the kept set keeps everything
    assert len(_nms(M, b[k], 0.1)) == len(k)
