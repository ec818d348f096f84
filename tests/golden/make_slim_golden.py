"""Generates tests/golden/slim_*.npz from the reference's own python:
  liso.slim.model.raft_code.corr.CorrBlock (+ bilinear_sampler), liso.slim.model.extractor.SmallEncoder,
  liso.slim.model.update.SmallUpdateBlock, liso.slim.model.raft_mod.RAFT.predict_single_flow_map_and_classes,
  liso.slim.model.head_decoder.HeadDecoder.concat2network_output
Weights are NOT stored (2.4 M parameters): modules are built under torch.manual_seed and the product's mirrors, built
in the same order under the same seed, must reproduce the stored state_dict checksum before outputs are compared.
mmcv/mmdet3d (pillar encoder, absent) and munch are stubbed with empty containers; RAFT is instantiated without its
`pp_layer` (object.__new__) because only predict_single_flow_map_and_classes is exercised.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_slim_golden.py
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
for name in ("mmcv", "mmcv.ops", "mmcv.runner", "mmcv.cnn", "mmdet3d", "mmdet3d.models", "mmdet3d.models.middle_encoders",
             "mmdet3d.models.middle_encoders.pillar_scatter", "mmdet3d.models.voxel_encoders",
             "mmdet3d.models.voxel_encoders.pillar_encoder"):
    sys.modules[name] = MagicMock()
munch = types.ModuleType("munch")
munch.Munch = type("Munch", (dict,), {"__getattr__": dict.__getitem__, "__setattr__": dict.__setitem__})
sys.modules["munch"] = munch


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def cfg(d):
    return _Cfg({k: cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values())), float(sum((v.double() ** 2).sum() for v in sd.values()))


def main():
    from liso.slim.model.extractor import SmallEncoder
    from liso.slim.model.head_decoder import HeadDecoder
    from liso.slim.model.raft_code.corr import CorrBlock
    from liso.slim.model.raft_mod import RAFT
    from liso.slim.model.update import SmallUpdateBlock

    out = {}
    # ---------------- CorrBlock: lookup + gradients w.r.t. both feature maps ----------------
    g = torch.Generator().manual_seed(0)
    B, D, h, w = 2, 128, 16, 16
    f1 = torch.randn(B, D, h, w, generator=g).requires_grad_(True)
    f2 = torch.randn(B, D, h, w, generator=g).requires_grad_(True)
    from liso.slim.model.raft_code.utils import coords_grid
    coords = coords_grid(B, h, w, "cpu") + torch.randn(B, 2, h, w, generator=g) * 2.5
    coords[0, :, 0, 0] = torch.tensor([-3.7, 20.2])   # far outside: zero padding
    coords[0, :, 1, 1] = torch.tensor([5.0, 7.0])     # exactly integral
    cb = CorrBlock(f1, f2, num_levels=4, radius=3)
    look = cb(coords)
    go = torch.randn(look.shape, generator=g)
    (look * go).sum().backward()
    out.update(corr_f1=f1.detach().numpy(), corr_f2=f2.detach().numpy(), corr_coords=coords.numpy(), corr_out=look.detach().numpy(),
               corr_go=go.numpy(), corr_gf1=f1.grad.numpy(), corr_gf2=f2.grad.numpy())

    # ---------------- encoders + update block + the 6-iteration RAFT loop ----------------
    slim_cfg = cfg({"model": {"num_iters": 6, "feature_downsampling_factor": 8, "flow_maps_archi": "single",
                              "predict_weight_for_static_aggregation": False,
                              "corr_cfg": {"module": "all", "num_levels": 4, "search_radius": 3},
                              "point_pillars": {"nbr_point_feats": 64}}})
    torch.manual_seed(1234)
    fnet = SmallEncoder(output_dim=128, norm_fn="instance_affine", dropout=0)
    cnet = SmallEncoder(output_dim=160, norm_fn="none", dropout=0)
    ub = SmallUpdateBlock(cfg=slim_cfg, filters=96)
    with torch.no_grad():  # non-trivial affine instance-norm parameters
        for m in fnet.modules():
            if isinstance(m, torch.nn.InstanceNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.2, 0.2)
    raft = object.__new__(RAFT)
    torch.nn.Module.__init__(raft)
    raft.slim_cfg, raft.cnet, raft.update_block = slim_cfg, cnet, ub
    raft.hidden_dim, raft.context_dim = 96, 64
    raft.bev_rows_res_meters_per_fs_pixel = raft.bev_cols_res_meters_per_fs_pixel = 40.0 / 128
    dec = HeadDecoder(slim_cfg, name="fw", bev_extent=None)
    gi = torch.Generator().manual_seed(5)
    img0 = torch.randn(1, 64, 128, 128, generator=gi) * (torch.rand(1, 1, 128, 128, generator=gi) > 0.8)
    img1 = torch.roll(img0, shifts=(3, -2), dims=(2, 3)) + 0.05 * torch.randn(1, 64, 128, 128, generator=gi)
    fmap0, fmap1 = fnet(img0), fnet(img1)
    preds = raft.predict_single_flow_map_and_classes(img0, fmap0, fmap1, dec)
    wts = [torch.randn(preds[0].shape, generator=gi) for _ in preds]
    sum((p * wt).sum() for p, wt in zip(preds, wts)).backward()
    sd = {"fnet." + k: v for k, v in fnet.state_dict().items()}
    sd.update({"cnet." + k: v for k, v in cnet.state_dict().items()})
    sd.update({"ub." + k: v for k, v in ub.state_dict().items()})
    # inputs are regenerated from the seed by the tests (same generator call sequence), not stored
    out.update(raft_checksum=np.asarray(checksum(sd)), raft_fmap0=fmap0.detach().numpy(),
               raft_pred_last=preds[-1].detach().numpy()[:, ::2, ::2], raft_pred_first=preds[0].detach().numpy()[:, ::4, ::4],
               raft_pred_means=np.asarray([float(p.mean()) for p in preds]),
               raft_loss_w_seed=np.asarray(5),
               raft_g_fnet_conv1=fnet.conv1.weight.grad.numpy(), raft_g_cnet_conv2=cnet.conv2.weight.grad.numpy(),
               raft_g_gru_convz=ub.gru.convz.weight.grad.numpy()[:, ::8], raft_g_flow_head=ub.static_flow_head.conv2.weight.grad.numpy(),
               raft_g_corr_conv=ub.motion_encoder.conv_stat_corr1.weight.grad.numpy()[..., 0, 0])
    np.savez_compressed(os.path.join(HERE, "slim_reference.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()
