"""Generates tests/golden/pillars_*.npz by running the reference's own python for the pillar path:
  VoxelGenerator / points_to_voxel   mmdetection3d/mmdet3d/core/voxel/voxel_generator.py
  PillarFeatureNet, PFNLayer         mmdetection3d/mmdet3d/models/voxel_encoders/{pillar_encoder,utils}.py
  PointPillarsScatter                mmdetection3d/mmdet3d/models/middle_encoders/pillar_scatter.py
glued exactly like liso/networks/pcl_to_feature_grid/pcl_to_feature_grid.py:58-102 (coors swap :73, batch pad :79-83).
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_pillars_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from oracle import pillars as OP  # noqa: E402  (only for the synthetic input generator + geometry helper)


def run_reference(pcls, grid, bev_range, z_cut, seed, training):
    utils, enc, sc, vg = ref_import.load_mmdet3d_pillar_modules()
    pc_range, voxel_size = OP.pillar_geometry((bev_range, bev_range), (grid, grid), z_cut)
    C = pcls[0].shape[1]
    torch.manual_seed(seed)
    pfn = enc.PillarFeatureNet(in_channels=C, feat_channels=[64], with_distance=False, voxel_size=voxel_size,
                               norm_cfg={"type": "BN1d", "eps": 0.001, "momentum": 0.01}, point_cloud_range=pc_range)
    # non-trivial BN affine + running stats so that eval mode is exercised too
    with torch.no_grad():
        pfn.pfn_layers[0].norm.weight.uniform_(0.5, 1.5)
        pfn.pfn_layers[0].norm.bias.uniform_(-0.5, 0.5)
        pfn.pfn_layers[0].norm.running_mean.uniform_(-1, 1)
        pfn.pfn_layers[0].norm.running_var.uniform_(0.5, 2.0)
    init = {k: v.detach().clone().numpy() for k, v in pfn.state_dict().items()}
    pfn.train(training)
    scatter = sc.PointPillarsScatter(in_channels=64, output_shape=(grid, grid))
    occ_scatter = sc.PointPillarsScatter(in_channels=1, output_shape=(grid, grid))
    voxels, coors, nums = [], [], []
    for b, p in enumerate(pcls):
        v, c, n = vg.points_to_voxel(p, voxel_size, pc_range, max_points=20, reverse_index=True, max_voxels=40000)
        c = c[:, [0, 2, 1]]                                     # pcl_to_feature_grid.py:73
        coors.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))  # :79-83
        voxels.append(v)
        nums.append(n)
    voxels, coors, nums = np.concatenate(voxels), np.concatenate(coors), np.concatenate(nums)
    vt, ct, nt = torch.from_numpy(voxels.copy()), torch.from_numpy(coors), torch.from_numpy(nums)
    feat = pfn(vt, nt, ct)
    bev = scatter(feat, ct, len(pcls))
    occ = occ_scatter(torch.ones_like(feat[:, [0]]), ct, len(pcls))
    out = {"voxels": voxels, "coors": coors, "num_points": nums, "bev": bev.detach().numpy(),
           "occupancy": occ.detach().numpy(), "voxel_features": feat.detach().numpy()}
    if training:
        # a fixed upstream gradient -> reference grads of the three trainable tensors + updated running stats
        g = torch.from_numpy(np.random.default_rng(seed + 1).standard_normal(bev.shape).astype(np.float32))
        (bev * g).sum().backward()
        lyr = pfn.pfn_layers[0]
        out.update(grad_out=g.numpy(), grad_weight=lyr.linear.weight.grad.numpy(), grad_gamma=lyr.norm.weight.grad.numpy(),
                   grad_beta=lyr.norm.bias.grad.numpy(), running_mean_after=lyr.norm.running_mean.numpy().copy(),
                   running_var_after=lyr.norm.running_var.numpy().copy())
    out.update({"init_" + k.replace(".", "__"): v for k, v in init.items()})
    return out


def main():
    cases = [
        # name, grid, range, z_cut, [n points per sample], channels, seed, training
        ("train_g64_b2", 64, 20.0, 10.0, [1500, 900], 4, 0, True),
        ("eval_g64_b1", 64, 20.0, 10.0, [1200], 4, 1, False),
        ("train_g512_b1", 512, 100.0, 10.0, [6000], 4, 2, True),
        ("train_g32_c3", 32, 10.0, 5.0, [700, 50, 300], 3, 3, True),
    ]
    for name, grid, rng, zc, ns, C, seed, training in cases:
        pcls = [OP.synthetic_cloud(n, seed * 10 + i, rng, C) for i, n in enumerate(ns)]
        out = run_reference(pcls, grid, rng, zc, seed, training)
        meta = dict(grid=grid, bev_range=rng, z_cut=zc, training=training, n_channels=C)
        keep = {k: v for k, v in out.items() if k not in ("voxels",)}  # voxels are implied by points + point order
        if True:
            # store the big canvas sparsely
            nz = np.nonzero(out["occupancy"][:, 0])
            keep["bev_nz_index"] = np.stack(nz, 1).astype(np.int32)
            keep["bev_nz_values"] = out["bev"][nz[0], :, nz[1], nz[2]]
            del keep["bev"], keep["occupancy"]
            if training:
                keep["grad_out_nz_values"] = out["grad_out"][nz[0], :, nz[1], nz[2]]
                del keep["grad_out"]
        np.savez_compressed(os.path.join(HERE, f"pillars_{name}.npz"), **keep,
                            **{f"pcl_{i}": p for i, p in enumerate(pcls)}, **{f"meta_{k}": np.asarray(v) for k, v in meta.items()})
        print(name, "P =", len(out["num_points"]), "max pts", out["num_points"].max())

    # the reference's own known-answer test for the CPU voxeliser (tests/test_models/test_voxel_encoder/
    # test_voxel_generator.py:8-22) is re-stated as data in tests/test_oracle_pillars.py, nothing to generate.


if __name__ == "__main__":
    main()
