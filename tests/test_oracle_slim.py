"""Pins oracle/slim_step.py::cpu_port -- the CPU restatement the full-size SLIM parity tests use as their checker
(tests/test_gpu_parity_full_size.py) -- to the reference's own outputs: tests/golden/slim_reference.npz, written by
liso/slim/model/{extractor,update,raft_mod,raft_code/corr}.py (tests/golden/make_slim_golden.py).  Under cpu_port() the
correlation is the explicit all-pairs volume + avg_pool2d + grid_sample of raft_code/corr.py:6-46 and every convolution /
normalisation runs through ATen on the host, so this is the reference's formulation end to end; the weights are rebuilt from the
seed and must reproduce the fixture's state_dict checksum first."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden", "slim_reference.npz")


def _rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def build_raft(seed=1234, grid=128, bev_range_m=40.0):
    """the modules of tests/golden/make_slim_golden.py, constructed in the same order under the same seed"""
    from liso_amd.slim.model.extractor import SmallEncoder
    from liso_amd.slim.model.head_decoder import HeadDecoder
    from liso_amd.slim.model.raft_mod import RAFT
    from liso_amd.slim.model.update import SmallUpdateBlock
    from liso_amd.utils.config import default_cfg

    cfg = default_cfg(grid=grid, bev_range_m=bev_range_m)
    torch.manual_seed(seed)
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
    raft.bev_rows_res_meters_per_fs_pixel = raft.bev_cols_res_meters_per_fs_pixel = bev_range_m / grid
    return fnet, cnet, ub, raft, HeadDecoder(cfg.SLIM, name="fw", bev_extent=None)


def test_cpu_port_reproduces_the_reference_raft_loop():
    from oracle.slim_step import cpu_port

    g = np.load(G)
    fnet, cnet, ub, raft, dec = build_raft()
    sd = {"fnet." + k: v for k, v in fnet.state_dict().items()}
    sd.update({"cnet." + k: v for k, v in cnet.state_dict().items()})
    sd.update({"ub." + k: v for k, v in ub.state_dict().items()})
    chk = (float(sum(v.double().abs().sum() for v in sd.values())), float(sum((v.double() ** 2).sum() for v in sd.values())))
    assert np.allclose(chk, g["raft_checksum"], rtol=1e-12), "seeded construction no longer reproduces the reference weights"
    gi = torch.Generator().manual_seed(5)
    img0 = torch.randn(1, 64, 128, 128, generator=gi) * (torch.rand(1, 1, 128, 128, generator=gi) > 0.8)
    img1 = torch.roll(img0, shifts=(3, -2), dims=(2, 3)) + 0.05 * torch.randn(1, 64, 128, 128, generator=gi)
    torch.set_num_threads(min(os.cpu_count() or 1, 8))
    with cpu_port():
        fmap0, fmap1 = fnet(img0), fnet(img1)
        preds = raft.predict_single_flow_map_and_classes(img0, fmap0, fmap1, dec)
        wts = [torch.randn(preds[0].shape, generator=gi) for _ in preds]
        sum((p * wt).sum() for p, wt in zip(preds, wts)).backward()
    # same arithmetic as the reference (ATen on the host): agreement to fp32 summation order
    assert _rel(fmap0, g["raft_fmap0"]) < 1e-5
    assert len(preds) == 6 and preds[0].shape == (1, 128, 128, 8)
    assert _rel(preds[0][:, ::4, ::4], g["raft_pred_first"]) < 1e-4
    assert _rel(preds[-1][:, ::2, ::2], g["raft_pred_last"]) < 1e-4
    assert np.allclose([float(p.mean()) for p in preds], g["raft_pred_means"], rtol=1e-4, atol=1e-6)
    assert _rel(cnet.conv2.weight.grad, g["raft_g_cnet_conv2"]) < 1e-3
    assert _rel(ub.gru.convz.weight.grad[:, ::8], g["raft_g_gru_convz"]) < 1e-3
    assert _rel(ub.static_flow_head.conv2.weight.grad, g["raft_g_flow_head"]) < 1e-3
    assert _rel(ub.motion_encoder.conv_stat_corr1.weight.grad[..., 0, 0], g["raft_g_corr_conv"]) < 1e-3


def test_cpu_port_correlation_lookup_reproduces_the_reference_corrblock():
    """the explicit-volume CorrBlock of cpu_port() against the reference CorrBlock's lookup and feature gradients (fixture keys corr_*)"""
    from oracle.slim_step import _CpuCorrBlock

    g = np.load(G)
    f1 = torch.from_numpy(g["corr_f1"]).requires_grad_(True)
    f2 = torch.from_numpy(g["corr_f2"]).requires_grad_(True)
    look = _CpuCorrBlock(f1, f2, num_levels=4, radius=3)(torch.from_numpy(g["corr_coords"]))
    assert _rel(look, g["corr_out"]) < 1e-5
    (look * torch.from_numpy(g["corr_go"])).sum().backward()
    assert _rel(f1.grad, g["corr_gf1"]) < 1e-5 and _rel(f2.grad, g["corr_gf2"]) < 1e-5
