"""Generates tests/golden/slim_loss_reference.npz from the reference's own python:
  liso.slim.model.head_decoder.HeadDecoder.forward  and
  liso.slim.slim_loss.slim_loss_adaptor.selfsupervisedSlimSingleScaleLoss
for (a) the default SLIM loss configuration and (b) the `slim_simple_knn_training` overlay.
Stubs: munch.Munch (attribute dict), pynanoflann.KDTree -> scipy.spatial.cKDTree (both are EXACT nearest-neighbour
searches; only distances enter the loss), `Tensor.cuda()` -> identity (the reference hard-codes `.cuda()`,
slim_loss_adaptor.py:145), shapely/mmcv empty modules.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_slim_loss_golden.py
"""
import copy
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
             "mmdet3d.models.voxel_encoders.pillar_encoder", "shapely", "shapely.affinity", "shapely.geometry"):
    sys.modules[name] = MagicMock()
munch = types.ModuleType("munch")


class Munch(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


munch.Munch = Munch
sys.modules["munch"] = munch
nf = types.ModuleType("pynanoflann")


class KDTree:
    def __init__(self, n_neighbors=1, metric="L2", leaf_size=20):
        self.k = n_neighbors

    def fit(self, index):
        from scipy.spatial import cKDTree
        self.tree = cKDTree(index)

    def kneighbors(self, queries):
        d, i = self.tree.query(queries, k=self.k)
        return d.reshape(len(queries), self.k), i.reshape(len(queries), self.k).astype(np.uint64)


nf.KDTree = KDTree
sys.modules["pynanoflann"] = nf
torch.Tensor.cuda = lambda self, *a, **k: self


def cfg(d):
    return Munch({k: cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


BASE = {
    "model": {"predict_weight_for_static_aggregation": False, "use_static_aggr_flow_for_aggr_flow": False,
              "dynamic_flow_is_non_rigid_flow": False, "u_net": {"final_scale": 1},
              "output_modification": {"disappearing_logit": False, "static_logit": "net", "dynamic_logit": "net",
                                      "ground_logit": False, "dynamic_flow": "net", "static_flow": "net",
                                      "dynamic_flow_grad_scale": 1.0}},
    "losses": {"unsupervised": {
        "fw_bw_static_trafo_penalty_factor": 1.0, "knn_loss_penalty_factor": 1.0,
        "artificial_labels": {"use_static_aggr_flow": True, "cross_entropy_penalty": 0.0, "weight_mode": "constant",
                              "gauss_widths": None, "knn_mode": "point"},
        "knn_on_dynamic_penalty": 0.0, "knn_on_static_penalty": 0.0, "knn_dist_measure": "point",
        "knn_loss": {"L1_delta": 0.0, "drop_outliers__perc": 0.0, "fov_mode": "mask_close_fov",
                     "range_based_weights": {"slope_sign": -1.0, "weight_slope": 0.0, "weight_at_range_0": 0.0,
                                             "max_weight_clip_at": 100.0, "min_weight_clip_at": 1.0}},
        "opposite_flow_penalty_factor": 0.0, "static_flow_penalty_factor": 1.0,
        "temporal_cls_consistency_penalty_factor": 0.0, "use_epsilon_for_weighted_pc_alignment": False}},
}


def main():
    from liso.slim.model.head_decoder import HeadDecoder
    from liso.slim.slim_loss.movavg_cls_threshold import MovingAverageThreshold
    from liso.slim.slim_loss.slim_loss_adaptor import selfsupervisedSlimSingleScaleLoss

    g = torch.Generator().manual_seed(0)
    B, N, G, R = 1, 3000, 32, 20.0
    ext = np.array([-R / 2, -R / 2, R / 2, R / 2])
    pc1 = torch.cat([torch.rand(B, N, 2, generator=g) * (R - 0.5) - (R - 0.5) / 2, torch.rand(B, N, 1, generator=g) * 2 - 1.5,
                     torch.rand(B, N, 1, generator=g)], -1)
    th, tx = 0.02, 0.4
    Rm = torch.tensor([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]], dtype=torch.float32)
    pc2 = pc1.clone()
    pc2[..., :2] = pc1[..., :2] @ Rm.T + torch.tensor([tx, 0.05]) + 0.02 * torch.randn(B, N, 2, generator=g)
    pc2[..., :2] = pc2[..., :2].clamp(-R / 2 + 0.01, R / 2 - 0.01)
    valid = torch.ones(B, N, dtype=torch.bool)
    coors = lambda pc: ((pc[..., :2] + R / 2) / R * G).to(torch.int32)
    odom = torch.eye(4, dtype=torch.float64)[None].clone()
    odom[0, :2, :2] = Rm.double(); odom[0, 0, 3] = tx; odom[0, 1, 3] = 0.05
    inv_odom = torch.linalg.inv(odom)

    def filled(c):
        m = torch.zeros(B, G, G, dtype=torch.bool)
        m[0, c[0, :, 0].long(), c[0, :, 1].long()] = True
        return m

    out = dict(pc1=pc1.numpy(), pc2=pc2.numpy(), odom=odom.numpy())
    for tag, overlay in (("default", {}), ("simple_knn", {"static_logit": True, "dynamic_logit": False, "ground_logit": False,
                                                            "dynamic_flow": "zero"})):
        c = copy.deepcopy(BASE)
        c["model"]["output_modification"].update(overlay)
        if tag == "simple_knn":
            c["losses"]["unsupervised"].update(fw_bw_static_trafo_penalty_factor=0.0, static_flow_penalty_factor=0.0)
        c = cfg(c)
        dec_fw, dec_bw = HeadDecoder(c, "fw", ext), HeadDecoder(c, "bw", ext)
        gn = torch.Generator().manual_seed(7)
        net_fw = (torch.randn(B, G, G, 8, generator=gn) * 0.5).requires_grad_(True)
        net_bw = (torch.randn(B, G, G, 8, generator=gn) * 0.5).requires_grad_(True)
        thr = MovingAverageThreshold(num_train_samples=100, num_moving=621013971, num_still=None)
        common = dict(summaries={"writer": None}, gt_flow_bev=None, ohe_gt_stat_dyn_ground_label_bev_map=None,
                      dynamic_flow_is_non_rigid_flow=False)
        pfw = dec_fw(net_fw, thr.value(), pc=pc1.clone(), pointwise_voxel_coordinates=coors(pc1), pointwise_valid_mask=valid,
                     filled_pillar_mask=filled(coors(pc1)), odom=odom, inv_odom=inv_odom, **common)
        pbw = dec_bw(net_bw, thr.value(), pc=pc2.clone(), pointwise_voxel_coordinates=coors(pc2), pointwise_valid_mask=valid,
                     filled_pillar_mask=filled(coors(pc2)), odom=inv_odom, inv_odom=odom, **common)
        metrics = {}
        loss = selfsupervisedSlimSingleScaleLoss(pc1=pc1.clone(), valid_mask_pc1=valid, pc2=pc2.clone(), valid_mask_pc2=valid,
                                                 pred_fw=pfw, pred_bw=pbw, moving_thresh_module=thr,
                                                 loss_cfg=c.losses.unsupervised, model_cfg=c.model, bev_extent=ext,
                                                 metrics_collector=metrics)
        loss.backward()
        out.update({f"{tag}_net_fw": net_fw.detach().numpy(), f"{tag}_net_bw": net_bw.detach().numpy(),
                    f"{tag}_loss": loss.detach().numpy(), f"{tag}_g_fw": net_fw.grad.numpy(), f"{tag}_g_bw": net_bw.grad.numpy(),
                    f"{tag}_agg_flow_fw": pfw.aggregated_flow.detach().numpy(), f"{tag}_staticness_fw": pfw.staticness.detach().numpy(),
                    f"{tag}_static_aggr_flow_fw": pfw.static_aggr_flow.detach().numpy(),
                    f"{tag}_T_fw": pfw.static_aggr_trafo.detach().numpy(), f"{tag}_is_static_fw": pfw.is_static.numpy(),
                    f"{tag}_dense_agg_fw": pfw.dense_maps.aggregated_flow.detach().numpy()})
        print(tag, "loss", float(loss), {k: float(v) for k, v in metrics.items()})
    np.savez_compressed(os.path.join(HERE, "slim_loss_reference.npz"), **out)


if __name__ == "__main__":
    main()
