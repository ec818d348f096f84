"""TEST INFRASTRUCTURE ONLY: the detector train step (forward + backward) on the host CPU, assembled from the oracle
restatements (oracle/pillars.py + oracle/detector.py).  Used by tests/ as the checker and by bench.py's cpu_baseline
leg as the timed CPU port ("kind": "port")."""
import time

import torch

from . import detector as OD
from . import pillars as OP

HEADS = ("pos", "dims", "rot", "probs")


def detector_forward_loss(sd, pcls, targets, grid, bev_range_m, z_cut=10.0, z_prior=(-1.5, -0.5), training=True,
                          dtype=torch.float32):
    """sd: BoxLearner state_dict (CPU tensors of `dtype`; the ones that need grads must already require them).
    dtype=torch.float64 gives the high-precision reference used to bound fp32 rounding noise in the parity tests."""
    pre = "model.pfn.pts_voxel_encoder.pfn_layers.0."
    targets = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in targets.items()}
    bev, occ, _ = OP.pillar_forward([p.numpy() if torch.is_tensor(p) else p for p in pcls], sd[pre + "linear.weight"],
                                    sd[pre + "norm.weight"], sd[pre + "norm.bias"], sd[pre + "norm.running_mean"],
                                    sd[pre + "norm.running_var"], training, (bev_range_m, bev_range_m), (grid, grid), z_cut,
                                    dtype=dtype)
    rsd = {k[len("model.rpn."):]: v for k, v in sd.items() if k.startswith("model.rpn.")}
    hsd = {k[len("model.center_head."):]: v for k, v in sd.items() if k.startswith("model.center_head.")}
    feat = OD.rpn_forward(rsd, bev, [3, 5, 5], [2, 2, 2], [0.5, 1, 2], training)
    pred = OD.center_head_forward(hsd, feat, HEADS, training)
    raw = {k: v.permute(0, 2, 3, 1) for k, v in pred.items()}
    dec, act = OD.decode(raw, sd["pillar_center_coors_m"], (bev_range_m, bev_range_m), z_prior[0], z_prior[1])
    gt = {k: targets[k] for k in HEADS}
    mask = targets["center_bool_mask"]
    losses = OD.centerpoint_loss(dec, act, gt, mask, torch.zeros_like(mask), torch.ones_like(gt["probs"]))
    total = sum(losses.values()) + 1e-4 * OD.rotation_regulariser(act)
    return total, raw, bev


def prepare_state(sd, dtype=torch.float32):
    """detach/clone a BoxLearner state_dict to CPU `dtype`; trainable tensors get requires_grad"""
    out = {}
    for k, v in sd.items():
        v = v.detach().cpu().clone()
        if v.dtype.is_floating_point:
            v = v.to(dtype)
            if "running" not in k and k != "pillar_center_coors_m":
                v.requires_grad_(True)
        out[k] = v
    return out


def timed_detector_step(sd, pcls, targets, grid, bev_range_m):
    """one forward+backward on the host cores; returns (seconds, loss)"""
    sd = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k and
              k != "pillar_center_coors_m" else v.detach().clone()) for k, v in sd.items()}
    t0 = time.perf_counter()
    total, _, _ = detector_forward_loss(sd, pcls, targets, grid, bev_range_m)
    total.backward()
    return time.perf_counter() - t0, float(total.detach())
