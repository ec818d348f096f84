"""Generates tests/golden/detector_*.npz from the reference's own torch modules (importable as they are):
  liso.networks.centerpoint.rpn.RPN, liso.networks.centerpoint.center_head.CenterHead,
  liso.kabsch.output_modification.output_modification, liso.losses.centerpoint_loss.centerpoint_loss
Reduced channel widths keep the fixture small; the architecture (layer counts, strides, kernel sizes) is the
CenterPoint-pillar one of liso/networks/simple_net/centerpoint_net.py:22-59.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_detector_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def cfg(d):
    return _Cfg({k: cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def main():
    from liso.kabsch.output_modification import output_modification
    from liso.losses.centerpoint_loss import centerpoint_loss
    from liso.networks.centerpoint.center_head import CenterHead
    from liso.networks.centerpoint.rpn import RPN

    torch.manual_seed(0)
    norm = {"affine": True, "track_running_stats": True}
    rpn = RPN(layer_nums=[3, 5, 5], ds_layer_strides=[2, 2, 2], ds_num_filters=[16, 32, 64],
              us_layer_strides=[0.5, 1, 2], us_num_filters=[32, 32, 32], num_input_features=16, norm_cfg=norm)
    heads = {"pos": (3, 2), "dims": (3, 2), "rot": (2, 2), "probs": (1, 2)}
    head = CenterHead(common_heads=heads, norm_cfg=norm, in_channels=96, stride=1, share_conv_channel=16)
    # SepHead's head_conv default is 64 (center_head.py:16) whatever share_conv_channel is
    with torch.no_grad():
        for m in list(rpn.modules()) + list(head.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.5, 0.5); m.running_var.uniform_(0.5, 1.5)
    sd = {"rpn." + k: v.clone() for k, v in rpn.state_dict().items()}
    sd.update({"head." + k: v.clone() for k, v in head.state_dict().items()})
    x = torch.randn(2, 16, 64, 64)
    out = {}
    for training in (True, False):
        rpn.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("rpn.")})
        head.load_state_dict({k[5:]: v for k, v in sd.items() if k.startswith("head.")})
        rpn.train(training); head.train(training)
        xi = x.clone().requires_grad_(True)
        feat = rpn(xi)
        pred = head(feat)
        tag = "train" if training else "eval"
        out[f"{tag}_feat"] = feat.detach().numpy()
        for k, v in pred.items():
            out[f"{tag}_{k}"] = v.detach().numpy()
        if training:
            loss = sum((v * torch.linspace(-1, 1, v.numel()).view_as(v)).sum() for v in pred.values())
            loss.backward()
            out["train_grad_x"] = xi.grad.numpy()
            out["train_grad_rpn_blocks_0_1_weight"] = rpn.blocks[0][1].weight.grad.numpy()
            out["train_grad_rpn_deblocks_2_0_weight"] = rpn.deblocks[2][0].weight.grad.numpy()
            out["train_grad_head_probs_3_bias"] = head.tasks[0].probs[3].bias.grad.numpy()
            out["train_rm_after_rpn_blocks_0_2"] = rpn.blocks[0][2].running_mean.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "detector_rpn_head.npz"), x=x.numpy(),
                        **{"sd_" + k.replace(".", "__"): v.numpy() for k, v in sd.items()}, **out)
    print("rpn/head fixture:", {k: v.shape for k, v in out.items() if "train_" in k and "grad" not in k})

    # ---- decode + loss ----
    B, H, W = 2, 16, 16
    g = torch.Generator().manual_seed(1)
    raw = {"pos": torch.randn(B, H, W, 3, generator=g), "dims": torch.randn(B, H, W, 3, generator=g),
           "rot": torch.randn(B, H, W, 2, generator=g), "probs": torch.randn(B, H, W, 1, generator=g)}
    raw = {k: v.requires_grad_(True) for k, v in raw.items()}
    box_cfg = cfg({"position_representation": {"method": "local_relative_offset", "num_box_pos_dims": 3,
                                               "box_z_pos_prior_min": -1.5, "box_z_pos_prior_max": -0.5},
                   "rotation_representation": {"method": "vector", "norm_vector_len": False},
                   "dimensions_representation": {"method": "predict_abs_size"},
                   "activations": {"pos": "tanh", "dims": "softplus", "rot": "none", "probs": "none"}})
    data_cfg = cfg({"bev_range_m": (40.0, 40.0)})
    from liso.utils.bev_utils import get_metric_voxel_center_coords
    centers = torch.from_numpy(get_metric_voxel_center_coords(40.0, 40.0, np.array([H, W])).astype(np.float32)[..., :2])
    act = {"pos": torch.tanh(raw["pos"]), "dims": torch.nn.functional.softplus(raw["dims"]), "rot": raw["rot"],
           "probs": raw["probs"]}  # simple_net.py:118-121 with simple_net_utils.py:8-14
    dec = output_modification({k: v.clone() for k, v in act.items()}, box_cfg, data_cfg, "boxes", centers)
    gt = {"probs": torch.rand(B, H, W, 1, generator=g), "rot": torch.randn(B, H, W, 2, generator=g),
          "dims": torch.rand(B, H, W, 3, generator=g) * 4, "pos": torch.randn(B, H, W, 3, generator=g) * 10}
    center_mask = torch.rand(B, H, W, generator=g) > 0.93
    ignore = torch.rand(B, H, W, generator=g) > 0.9
    gt["probs"][center_mask] = 1.0
    loss_cfg = cfg({"supervised": {"centermaps": {"confidence_target": "gaussian"}}})
    losses = centerpoint_loss(loss_cfg=loss_cfg, decoded_pred_box_maps=dec, raw_activated_pred_box_maps=act, gt_maps=gt,
                              gt_center_mask=center_mask, rotation_loss_weights_map=torch.ones_like(gt["probs"]),
                              box_prediction_cfg=box_cfg, ignore_region_is_true_mask=ignore)
    total = sum(losses.values())
    total.backward()
    fix = {"raw_" + k: v.detach().numpy() for k, v in raw.items()}
    fix.update({"grad_" + k: v.grad.numpy() for k, v in raw.items()})
    fix.update({"dec_" + k: v.detach().numpy() for k, v in dec.items()})
    fix.update({"gt_" + k: v.numpy() for k, v in gt.items()})
    fix.update({"loss_" + k.split("/")[-1]: v.detach().numpy() for k, v in losses.items()})
    np.savez_compressed(os.path.join(HERE, "detector_decode_loss.npz"), center_mask=center_mask.numpy(),
                        ignore=ignore.numpy(), centers=centers.numpy(), **fix)
    print("loss fixture:", {k: float(v) for k, v in losses.items()})


if __name__ == "__main__":
    main()
