"""Generates tests/golden/kabsch_*.npz from the reference's own python:
  liso.torch_symm_ortho.symmetric_orthogonalization (forward + its analytic backward)
  liso.slim.slim_loss.weighted_pc_alignment.weighted_pc_alignment
  liso.kabsch.kabsch_mask.KabschDecoder.get_kabsch_trafos_from_point_flow
shapely (absent) is stubbed with empty modules: only Shape.get_shapely_contour would use it, and nothing here calls it.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_kabsch_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
for name in ("shapely", "shapely.affinity", "shapely.geometry"):
    m = types.ModuleType(name)
    m.rotate = m.translate = m.Point = m.box = None
    sys.modules[name] = m


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def cfg(d):
    return _Cfg({k: cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def main():
    from liso.kabsch.kabsch_mask import KabschDecoder
    from liso.kabsch.shape_utils import Shape
    from liso.slim.slim_loss.weighted_pc_alignment import weighted_pc_alignment
    from liso.torch_symm_ortho import symmetric_orthogonalization

    g = torch.Generator().manual_seed(0)
    # --- symmetric orthogonalisation fwd/bwd, incl. a reflection (det < 0) and a badly scaled matrix ---
    A = torch.randn(12, 3, 3, generator=g, dtype=torch.float64)
    A[1] = A[1] * 1e-4
    A[2] = torch.diag(torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)) @ A[2]
    A[3] = A[3] + 5 * torch.eye(3, dtype=torch.float64)
    A = A.requires_grad_(True)
    R = symmetric_orthogonalization(A)
    G = torch.randn(12, 3, 3, generator=g, dtype=torch.float64)
    (R * G).sum().backward()
    out = {"so_A": A.detach().numpy(), "so_R": R.detach().numpy(), "so_G": G.numpy(), "so_gradA": A.grad.numpy()}

    # --- weighted_pc_alignment (3-D, SLIM static aggregation call shape) ---
    n = 4000
    p0 = torch.randn(n, 3, generator=g) * torch.tensor([20.0, 20.0, 1.0])
    th = 0.03
    Rt = torch.tensor([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], dtype=torch.float32)
    p1 = p0 @ Rt.T + torch.tensor([0.8, -0.1, 0.0]) + 0.05 * torch.randn(n, 3, generator=g)
    p1[:, 2] = p0[:, 2]
    w = torch.rand(n, generator=g).requires_grad_(True)
    T, nep = weighted_pc_alignment(p0, p1, w)
    GT = torch.randn(4, 4, generator=g, dtype=torch.float64)
    (T * GT).sum().backward()
    out.update(wpa_p0=p0.numpy(), wpa_p1=p1.numpy(), wpa_w=w.detach().numpy(), wpa_T=T.detach().numpy(), wpa_GT=GT.numpy(),
               wpa_grad_w=w.grad.numpy(), wpa_nep=np.asarray(bool(nep)))
    few = torch.zeros(n); few[:2] = 1.0
    T2, nep2 = weighted_pc_alignment(p0, p1, few)
    out.update(wpa_few_T=T2.numpy(), wpa_few_nep=np.asarray(bool(nep2)))

    # --- KabschDecoder on padded clouds ---
    c = cfg({"data": {"bev_range_m": (100.0, 100.0), "img_grid_size": (64, 64), "shapes": {"name": "boxes"}},
             "mask_rendering": {"softness_fun": "cauchy", "pred_sigmoid_slope": 15.0, "obj_dim_scale_buffer": 0.25},
             "svd_backend": "symm_ortho"})
    dec = KabschDecoder(c)
    B, N, S = 2, 6000, 7
    pts = torch.cat([torch.rand(B, N, 2, generator=g) * 80 - 40, torch.rand(B, N, 1, generator=g) * 3 - 2], dim=-1)
    pos = torch.cat([torch.rand(B, S, 2, generator=g) * 60 - 30, torch.full((B, S, 1), -0.8)], dim=-1)
    dims = torch.stack([torch.rand(B, S, generator=g) * 2 + 3.5, torch.rand(B, S, generator=g) + 1.5,
                        torch.rand(B, S, generator=g) * 0.5 + 1.4], dim=-1)
    rot = (torch.rand(B, S, 1, generator=g) * 2 - 1) * np.pi
    # put a cluster of points into every box and give it a rigid motion
    flow = torch.zeros(B, N, 3)
    flow[..., 0] = 0.4  # ego motion seen as background flow
    per = 150
    for b in range(B):
        for s in range(S):
            sl = slice(s * per, (s + 1) * per)
            loc = (torch.rand(per, 3, generator=g) - 0.5) * dims[b, s]
            cs, sn = torch.cos(rot[b, s, 0]), torch.sin(rot[b, s, 0])
            pts[b, sl, 0] = pos[b, s, 0] + cs * loc[:, 0] - sn * loc[:, 1]
            pts[b, sl, 1] = pos[b, s, 1] + sn * loc[:, 0] + cs * loc[:, 1]
            pts[b, sl, 2] = pos[b, s, 2] + loc[:, 2]
            v = 0.3 * (s + 1)
            flow[b, sl, 0], flow[b, sl, 1] = v * cs, v * sn
    flow += 0.01 * torch.randn(B, N, 3, generator=g)
    valid = torch.ones(B, N, dtype=torch.bool)
    valid[0, 5000:] = False
    pts_nan = pts.clone(); pts_nan[~valid] = float("nan")
    flow_nan = flow.clone(); flow_nan[~valid] = float("nan")
    boxes = Shape(pos=pos.clone(), dims=dims.clone(), rot=rot.clone(), probs=torch.ones(B, S, 1))
    fgT, fgw, fgc, bgT, bgc = dec.get_kabsch_trafos_from_point_flow(
        point_cloud_ta=pts_nan.clone(), valid_mask_ta=valid, pointwise_flow_ta_tb=flow_nan.clone(), pred_boxes_ta=boxes)
    out.update(kd_pts=pts_nan.numpy(), kd_flow=flow_nan.numpy(), kd_valid=valid.numpy(), kd_pos=pos.numpy(), kd_dims=dims.numpy(),
               kd_rot=rot.numpy(), kd_fgT=fgT.numpy(), kd_fgw_sum=fgw.sum(-1).numpy(), kd_fgw_sample=fgw[:, :, ::50].numpy(),
               kd_fgc=fgc.numpy(), kd_bgT=bgT.numpy(), kd_bgc=bgc.numpy())
    # empty-weight slot (far away tiny sigmoid box) exercises the epsilon rule with the sigmoid softness
    far = Shape(pos=torch.tensor([[[4000.0, 4000.0, 0.0]]]), dims=torch.tensor([[[1.0, 1.0, 1.0]]]), rot=torch.zeros(1, 1, 1),
                probs=torch.ones(1, 1, 1))
    fgT2, _, fgc2, bgT2, bgc2 = dec.get_kabsch_trafos_from_point_flow(
        point_cloud_ta=pts_nan[:1].clone(), valid_mask_ta=valid[:1], pointwise_flow_ta_tb=flow_nan[:1].clone(),
        pred_boxes_ta=far, softness_func=torch.sigmoid)
    out.update(kd_far_fgT=fgT2.numpy(), kd_far_fgc=fgc2.numpy(), kd_far_bgT=bgT2.numpy(), kd_far_bgc=bgc2.numpy())
    # --- KabschDecoder.forward without points: masks rendered on the BEV grid's pillar centres (kabsch_mask.py:274-276) ---
    grid_w, _ = dec(boxes, obj_dim_scale=1.25)
    grid_ws, _ = dec(boxes, softness_func=torch.sigmoid, sigmoid_slope=7.0)
    out.update(kd_grid_w=grid_w.numpy(), kd_grid_w_sigmoid=grid_ws.numpy())
    np.savez_compressed(os.path.join(HERE, "kabsch_reference.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print("fg cum", fgc[0, :3], "far cum", fgc2, "R22", fgT[0, :, 2, 2])


if __name__ == "__main__":
    main()
