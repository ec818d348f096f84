"""GPU parity: on-the-fly correlation lookup kernel and the SLIM RAFT loop vs fixtures from the reference's modules."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REL = 1e-3


def _g():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "slim_reference.npz"))


def _rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def test_corr_lookup_matches_reference_volume_path():
    from liso_amd.slim.model.raft_code.corr import CorrBlock

    g = _g()
    f1 = torch.from_numpy(g["corr_f1"]).cuda().requires_grad_(True)
    f2 = torch.from_numpy(g["corr_f2"]).cuda().requires_grad_(True)
    cb = CorrBlock(f1, f2, num_levels=4, radius=3)
    out = cb(torch.from_numpy(g["corr_coords"]).cuda())
    assert out.shape == (2, 196, 16, 16) and out.dtype == torch.float32
    assert _rel(out, g["corr_out"]) < 1e-4
    (out * torch.from_numpy(g["corr_go"]).cuda()).sum().backward()
    assert _rel(f1.grad, g["corr_gf1"]) < 1e-4 and _rel(f2.grad, g["corr_gf2"]) < 1e-4


@pytest.mark.parametrize("h,w,B", [(64, 64, 1), (128, 128, 2), (24, 40, 3)])
def test_corr_lookup_vs_explicit_volume(h, w, B):
    """BASELINE sizes (64x64 = 512^2 BEV / 8, 128x128 = 1024^2 / 8) against the explicit all-pairs volume + pooling +
    grid_sample formulation of the reference (corr.py:6-46) evaluated with torch ops on the same device."""
    import torch.nn.functional as F

    from liso_amd.slim.model.raft_code.corr import CorrBlock
    from liso_amd.slim.model.raft_code.utils import bilinear_sampler, coords_grid

    torch.manual_seed(h * w + B)
    f1 = torch.randn(B, 128, h, w, device="cuda")
    f2 = torch.randn(B, 128, h, w, device="cuda")
    coords = coords_grid(B, h, w, "cuda") + torch.randn(B, 2, h, w, device="cuda") * 3
    out = CorrBlock(f1, f2, 4, 3)(coords)
    vol = CorrBlock.corr(f1, f2).reshape(B * h * w, 1, h, w)
    ref = []
    c = coords.permute(0, 2, 3, 1)
    for i in range(4):
        d = torch.linspace(-3, 3, 7, device="cuda")
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
        ref.append(bilinear_sampler(vol, c.reshape(B * h * w, 1, 1, 2) / 2 ** i + delta.view(1, 7, 7, 2)).view(B, h, w, -1))
        vol = F.avg_pool2d(vol, 2, stride=2)
    ref = torch.cat(ref, dim=-1).permute(0, 3, 1, 2)
    assert _rel(out, ref.cpu().numpy()) < 1e-4


def _build(seed=1234):
    from liso_amd.slim.model.extractor import SmallEncoder
    from liso_amd.slim.model.head_decoder import HeadDecoder
    from liso_amd.slim.model.raft_mod import RAFT
    from liso_amd.slim.model.update import SmallUpdateBlock
    from liso_amd.utils.config import default_cfg

    cfg = default_cfg(grid=128, bev_range_m=40.0)
    torch.manual_seed(seed)  # same construction order as tests/golden/make_slim_golden.py
    fnet = SmallEncoder(output_dim=128, norm_fn="instance_affine", dropout=0)
    cnet = SmallEncoder(output_dim=160, norm_fn="none", dropout=0)
    ub = SmallUpdateBlock(cfg=cfg.SLIM, filters=96)
    with torch.no_grad():
        for m in fnet.modules():
            if isinstance(m, torch.nn.InstanceNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.2, 0.2)
    raft = object.__new__(RAFT)
    torch.nn.Module.__init__(raft)
    raft.cfg, raft.slim_cfg, raft.cnet, raft.update_block = cfg, cfg.SLIM, cnet, ub
    raft.hidden_dim, raft.context_dim = 96, 64
    raft.bev_rows_res_meters_per_fs_pixel = raft.bev_cols_res_meters_per_fs_pixel = 40.0 / 128
    return fnet, cnet, ub, raft, HeadDecoder(cfg.SLIM, name="fw", bev_extent=None)


def test_raft_loop_matches_reference():
    g = _g()
    fnet, cnet, ub, raft, dec = _build()
    sd = {"fnet." + k: v for k, v in fnet.state_dict().items()}
    sd.update({"cnet." + k: v for k, v in cnet.state_dict().items()})
    sd.update({"ub." + k: v for k, v in ub.state_dict().items()})
    chk = (float(sum(v.double().abs().sum() for v in sd.values())), float(sum((v.double() ** 2).sum() for v in sd.values())))
    assert np.allclose(chk, g["raft_checksum"], rtol=1e-12), "seeded construction no longer reproduces the reference weights"
    gi = torch.Generator().manual_seed(5)
    img0 = torch.randn(1, 64, 128, 128, generator=gi) * (torch.rand(1, 1, 128, 128, generator=gi) > 0.8)
    img1 = torch.roll(img0, shifts=(3, -2), dims=(2, 3)) + 0.05 * torch.randn(1, 64, 128, 128, generator=gi)
    for m in (fnet, cnet, ub):
        m.cuda()
    img0, img1 = img0.cuda(), img1.cuda()
    fmap0, fmap1 = fnet(img0), fnet(img1)
    assert _rel(fmap0, g["raft_fmap0"]) < REL
    preds = raft.predict_single_flow_map_and_classes(img0, fmap0, fmap1, dec)
    assert len(preds) == 6 and preds[0].shape == (1, 128, 128, 8)
    assert _rel(preds[0][:, ::4, ::4], g["raft_pred_first"]) < REL
    assert _rel(preds[-1][:, ::2, ::2], g["raft_pred_last"]) < REL
    assert np.allclose([float(p.mean()) for p in preds], g["raft_pred_means"], rtol=1e-3, atol=1e-5)
    wts = [torch.randn(preds[0].shape, generator=gi).cuda() for _ in preds]
    sum((p * wt).sum() for p, wt in zip(preds, wts)).backward()
    assert _rel(fnet.conv1.weight.grad, g["raft_g_fnet_conv1"]) < 5e-3
    assert _rel(cnet.conv2.weight.grad, g["raft_g_cnet_conv2"]) < 5e-3
    assert _rel(ub.gru.convz.weight.grad[:, ::8], g["raft_g_gru_convz"]) < 5e-3
    assert _rel(ub.static_flow_head.conv2.weight.grad, g["raft_g_flow_head"]) < 5e-3
    assert _rel(ub.motion_encoder.conv_stat_corr1.weight.grad[..., 0, 0], g["raft_g_corr_conv"]) < 5e-3
