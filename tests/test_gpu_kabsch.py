"""GPU parity: fused gfx950 Kabsch path and the 3x3 Jacobi symmetric orthogonalisation vs the reference fixtures and
the CPU oracle (tolerance 1e-3 relative, north_star)."""
import os

import numpy as np
import pytest
import torch

from oracle import kabsch as OK

pytestmark = pytest.mark.gpu
REL = 1e-3


def _g():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "kabsch_reference.npz"))


def _rel(a, b):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def _decoder():
    from liso_amd.kabsch.kabsch_mask import KabschDecoder
    from liso_amd.utils.config import default_cfg

    return KabschDecoder(default_cfg(grid=64, bev_range_m=100.0)).cuda()


def test_symm_ortho_matches_reference_fwd_bwd():
    from liso_amd.torch_symm_ortho import symmetric_orthogonalization

    g = _g()
    A = torch.from_numpy(g["so_A"]).cuda().requires_grad_(True)
    R = symmetric_orthogonalization(A)
    assert _rel(R, g["so_R"]) < 1e-9
    (R * torch.from_numpy(g["so_G"]).cuda()).sum().backward()
    assert _rel(A.grad, g["so_gradA"]) < 1e-7
    # batch shapes, fp32 inputs, orthogonality, rank-deficient input
    x = torch.randn(5, 7, 3, 3, device="cuda")
    r = symmetric_orthogonalization(x)
    assert r.shape == x.shape and r.dtype == torch.float32
    assert torch.allclose(r @ r.transpose(-1, -2), torch.eye(3, device="cuda").expand_as(r), atol=1e-5)
    flat = torch.zeros(3, 3, dtype=torch.float64, device="cuda")
    flat[:2, :2] = torch.tensor([[0.3, -0.9], [0.8, 0.2]], dtype=torch.float64)
    rf = symmetric_orthogonalization(flat[None])[0]
    assert abs(float(rf[2, 2]) - 1.0) < 1e-12 and torch.allclose(rf @ rf.T, torch.eye(3, dtype=torch.float64, device="cuda"))


def test_weighted_pc_alignment_matches_reference():
    from liso_amd.slim.slim_loss.weighted_pc_alignment import weighted_pc_alignment

    g = _g()
    w = torch.from_numpy(g["wpa_w"]).cuda().requires_grad_(True)
    T, nep = weighted_pc_alignment(torch.from_numpy(g["wpa_p0"]).cuda(), torch.from_numpy(g["wpa_p1"]).cuda(), w)
    assert T.dtype == torch.float64 and _rel(T, g["wpa_T"]) < 1e-5 and bool(nep) == bool(g["wpa_nep"])
    (T * torch.from_numpy(g["wpa_GT"]).cuda()).sum().backward()
    assert _rel(w.grad, g["wpa_grad_w"]) < REL
    few = torch.zeros(4000, device="cuda")
    few[:2] = 1.0
    T2, nep2 = weighted_pc_alignment(torch.from_numpy(g["wpa_p0"]).cuda(), torch.from_numpy(g["wpa_p1"]).cuda(), few)
    assert bool(nep2) and _rel(T2, g["wpa_few_T"]) < REL


def test_kabsch_decoder_matches_reference():
    from liso_amd.kabsch.shape_utils import Shape

    g = _g()
    dec = _decoder()
    S = g["kd_pos"].shape[1]
    boxes = Shape(pos=torch.from_numpy(g["kd_pos"]).cuda(), dims=torch.from_numpy(g["kd_dims"]).cuda(),
                  rot=torch.from_numpy(g["kd_rot"]).cuda(), probs=torch.ones(2, S, 1, device="cuda"))
    fgT, fgw, fgc, bgT, bgc = dec.get_kabsch_trafos_from_point_flow(
        point_cloud_ta=torch.from_numpy(g["kd_pts"]).cuda(), valid_mask_ta=torch.from_numpy(g["kd_valid"]).cuda(),
        pointwise_flow_ta_tb=torch.from_numpy(g["kd_flow"]).cuda(), pred_boxes_ta=boxes)
    assert fgT.dtype == torch.float64 and fgT.shape == (2, S, 4, 4) and bgT.shape == (2, 1, 4, 4)
    assert _rel(fgT, g["kd_fgT"]) < REL and _rel(bgT, g["kd_bgT"]) < REL
    assert _rel(fgc, g["kd_fgc"]) < REL and _rel(bgc, g["kd_bgc"]) < REL
    assert _rel(fgw.sum(-1), g["kd_fgw_sum"]) < REL and _rel(fgw[:, :, ::50], g["kd_fgw_sample"]) < REL
    far = Shape(pos=torch.tensor([[[4000.0, 4000.0, 0.0]]]).cuda(), dims=torch.tensor([[[1.0, 1.0, 1.0]]]).cuda(),
                rot=torch.zeros(1, 1, 1).cuda(), probs=torch.ones(1, 1, 1).cuda())
    fgT2, _, fgc2, bgT2, bgc2 = dec.get_kabsch_trafos_from_point_flow(
        point_cloud_ta=torch.from_numpy(g["kd_pts"][:1]).cuda(), valid_mask_ta=torch.from_numpy(g["kd_valid"][:1]).cuda(),
        pointwise_flow_ta_tb=torch.from_numpy(g["kd_flow"][:1]).cuda(), pred_boxes_ta=far, softness_func=torch.sigmoid)
    assert _rel(fgc2, g["kd_far_fgc"]) < REL and _rel(fgT2, g["kd_far_fgT"]) < REL and _rel(bgT2, g["kd_far_bgT"]) < REL


@pytest.mark.parametrize("B,N,S", [(1, 120000, 30), (2, 50000, 70), (1, 1000, 0), (3, 333, 5)])
def test_baseline_size_vs_oracle(B, N, S):
    from liso_amd.kabsch.shape_utils import Shape

    g = torch.Generator().manual_seed(N + S)
    pts = torch.cat([torch.rand(B, N, 2, generator=g) * 90 - 45, torch.rand(B, N, 1, generator=g) * 3 - 2], -1)
    flow = torch.randn(B, N, 3, generator=g) * 0.3
    valid = torch.rand(B, N, generator=g) > 0.1
    pos = torch.cat([torch.rand(B, S, 2, generator=g) * 70 - 35, torch.full((B, S, 1), -0.7)], -1)
    dims = torch.rand(B, S, 3, generator=g) * 2 + 1.5
    rot = (torch.rand(B, S, 1, generator=g) * 2 - 1) * 3.1
    T, cum, w = OK.kabsch_trafos(pos, dims, rot, pts, valid, flow)
    dec = _decoder()
    boxes = Shape(pos=pos.cuda(), dims=dims.cuda(), rot=rot.cuda(), probs=torch.ones(B, S, 1).cuda())
    nan_pts = pts.clone()
    nan_pts[~valid] = float("nan")
    fgT, fgw, fgc, bgT, bgc = dec.get_kabsch_trafos_from_point_flow(
        point_cloud_ta=nan_pts.cuda(), valid_mask_ta=valid.cuda(), pointwise_flow_ta_tb=flow.cuda(), pred_boxes_ta=boxes)
    if S:
        assert _rel(fgT, T[:, :S]) < REL and _rel(fgc, cum[:, :S]) < REL and _rel(fgw, w) < REL
    assert _rel(bgT, T[:, S:]) < REL and _rel(bgc, cum[:, S:]) < REL
    # size-independent property: a globally rigid flow is recovered by the background slot
    th, tx, ty = 0.02, 0.7, -0.3
    Rm = torch.tensor([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]], dtype=torch.float32)
    rigid = torch.cat([pts[..., :2] @ Rm.T + torch.tensor([tx, ty]) - pts[..., :2], torch.zeros(B, N, 1)], -1)
    _, _, _, bgT2, _ = dec.get_kabsch_trafos_from_point_flow(
        point_cloud_ta=nan_pts.cuda(), valid_mask_ta=valid.cuda(), pointwise_flow_ta_tb=rigid.cuda(), pred_boxes_ta=boxes)
    expect = torch.eye(4, dtype=torch.float64)
    expect[:2, :2], expect[0, 3], expect[1, 3] = Rm.double(), tx, ty
    assert torch.allclose(bgT2.cpu()[:, 0], expect.expand(B, 4, 4), atol=2e-4)


def test_weighted_moments_forward_backward_vs_torch_fp64():
    """include/liso_kabsch.h liso_weighted_moments_*: the fused reduction and its backward against the same sums
    written with torch fp64 ops (and autograd); bit-reproducible"""
    from liso_amd.slim.slim_loss.weighted_pc_alignment import _WeightedMoments

    g = torch.Generator().manual_seed(4)
    n = 120000
    x = (torch.rand(n, 3, generator=g) * 100 - 50).cuda().requires_grad_(True)
    y = (x.detach().cpu() + torch.randn(n, 3, generator=g) * 0.3).cuda().requires_grad_(True)
    w = torch.rand(n, generator=g).cuda()
    w[::5] = 0.0
    w.requires_grad_(True)
    go = torch.randn(16, generator=g, dtype=torch.float64).cuda()
    out = _WeightedMoments.apply(x, y, w)
    xd, yd, wd = x.double(), y.double(), w.double()
    ref = torch.cat([wd.sum()[None], (wd[:, None] * xd).sum(0), (wd[:, None] * yd).sum(0),
                     torch.einsum("n,na,nb->ab", wd, yd, xd).reshape(-1)])
    assert torch.allclose(out, ref, rtol=1e-12, atol=1e-9)
    assert torch.equal(out, _WeightedMoments.apply(x, y, w))
    ga = torch.autograd.grad((out * go).sum(), [x, y, w])
    gb = torch.autograd.grad((ref * go).sum(), [x, y, w])
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))


def test_bev_grid_mask_rendering_matches_reference():
    """KabschDecoder.forward without points (kabsch_mask.py:274-276): soft masks of every box slot on the BEV grid's pillar centres,
    [B,S,gx,gy] -- the reference's own output for both softness functions (fixture keys kd_grid_w*)"""
    from liso_amd.kabsch.shape_utils import Shape

    g = _g()
    dec = _decoder()
    boxes = Shape(pos=torch.from_numpy(g["kd_pos"]).cuda(), dims=torch.from_numpy(g["kd_dims"]).cuda(), rot=torch.from_numpy(g["kd_rot"]).cuda(),
                  probs=torch.ones(2, 7, 1, device="cuda"))
    w, logits = dec(boxes, obj_dim_scale=1.25)
    assert logits is None and w.shape == (2, 7, 64, 64)
    assert float((w.cpu() - torch.from_numpy(g["kd_grid_w"])).abs().max()) <= 1e-5
    ws, _ = dec(boxes, softness_func=torch.sigmoid, sigmoid_slope=7.0)
    assert float((ws.cpu() - torch.from_numpy(g["kd_grid_w_sigmoid"])).abs().max()) <= 1e-5
