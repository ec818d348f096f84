"""CPU: pins oracle/pillars.py against (a) the reference's own known-answer test of its CPU voxeliser and
(b) fixtures produced by the reference's python (tests/golden/make_pillars_golden.py)."""
import glob
import os

import numpy as np
import torch

from oracle import pillars as OP


def _load(f):
    g = np.load(f)
    pcls = [g[k] for k in sorted((k for k in g.files if k.startswith("pcl_")), key=lambda s: int(s[4:]))]
    meta = {k[5:]: g[k].item() for k in g.files if k.startswith("meta_")}
    return g, pcls, meta


def _dense(g, B, grid, key_vals="bev_nz_values"):
    idx = g["bev_nz_index"]
    out = np.zeros((B, 64, grid, grid), np.float32)
    out[idx[:, 0], :, idx[:, 1], idx[:, 2]] = g[key_vals]
    return out, idx


def test_reference_known_answer_voxel_generator():
    """mmdetection3d/tests/test_models/test_voxel_encoder/test_voxel_generator.py:8-22 restated as data."""
    np.random.seed(0)
    points = np.random.rand(1000, 4)
    voxels, coors, num, _ = OP.voxelize_hard(points, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], max_points=1000,
                                             max_voxels=20000)
    expected_coors = np.array([[7, 81, 1], [6, 81, 0], [7, 80, 1], [6, 81, 1], [7, 81, 0], [6, 80, 1], [7, 80, 0],
                               [6, 80, 0]])
    expected_num = np.array([120, 121, 127, 134, 115, 127, 125, 131])
    assert voxels.shape == (8, 1000, 4)
    assert np.all(coors == expected_coors)
    assert np.all(num == expected_num)


def test_oracle_matches_reference_fixtures(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "pillars_*.npz")))
    assert files
    for f in files:
        g, pcls, meta = _load(f)
        grid, rng, zc, training = meta["grid"], meta["bev_range"], meta["z_cut"], bool(meta["training"])
        w = torch.from_numpy(g["init_pfn_layers__0__linear__weight"]).clone().requires_grad_(True)
        gamma = torch.from_numpy(g["init_pfn_layers__0__norm__weight"]).clone().requires_grad_(True)
        beta = torch.from_numpy(g["init_pfn_layers__0__norm__bias"]).clone().requires_grad_(True)
        rm = torch.from_numpy(g["init_pfn_layers__0__norm__running_mean"]).clone()
        rv = torch.from_numpy(g["init_pfn_layers__0__norm__running_var"]).clone()
        bev, occ, (v, n, c, pi) = OP.pillar_forward(pcls, w, gamma, beta, rm, rv, training, (rng, rng), (grid, grid), zc)
        # integer products: bit-exact
        assert np.array_equal(c, g["coors"]), f
        assert np.array_equal(n, g["num_points"]), f
        ref_bev, idx = _dense(g, len(pcls), grid)
        assert np.array_equal(np.stack(np.nonzero(occ[:, 0].numpy()), 1), idx), f
        assert np.allclose(bev.detach().numpy(), ref_bev, rtol=1e-5, atol=1e-5), f
        if training:
            gout, _ = _dense(g, len(pcls), grid, "grad_out_nz_values")
            (bev * torch.from_numpy(gout)).sum().backward()
            for t, k in ((w, "grad_weight"), (gamma, "grad_gamma"), (beta, "grad_beta")):
                assert np.allclose(t.grad.numpy(), g[k], rtol=1e-4, atol=1e-4 * np.abs(g[k]).max()), (f, k)
            assert np.allclose(rm.numpy(), g["running_mean_after"], rtol=1e-5, atol=1e-6)
            assert np.allclose(rv.numpy(), g["running_var_after"], rtol=1e-5, atol=1e-6)


def test_voxelizer_edge_cases():
    pc_range, vs = OP.pillar_geometry((10.0, 10.0), (8, 8), 5.0)
    # empty cloud, all-outside cloud, boundary points
    for pts in (np.zeros((0, 4), np.float32), np.full((5, 4), 100.0, np.float32)):
        v, c, n, pi = OP.voxelize_hard(pts, vs, pc_range)
        assert len(n) == 0 and v.shape == (0, 20, 4)
    b = np.array([[-5.0, -5.0, 0, 0], [4.999999, 4.999999, 0, 0], [5.0, 0, 0, 0], [0, 0, 5.0, 0], [0, 0, -5.0, 0]], np.float32)
    v, c, n, pi = OP.voxelize_hard(b, vs, pc_range)
    assert [tuple(x) for x in c] == [(0, 0, 0), (0, 7, 7), (0, 4, 4)]  # x=5 and z=5 are outside, z=-5 is inside
    # > max_points in one voxel keeps the first 20 in point order; max_voxels drops late voxels
    many = np.zeros((50, 4), np.float32)
    many[:, 3] = np.arange(50)
    v, c, n, pi = OP.voxelize_hard(many, vs, pc_range)
    assert n.tolist() == [20] and pi[0].tolist() == list(range(20))
    r = np.random.default_rng(0)
    spread = np.concatenate([r.uniform(-5, 5, (300, 2)), np.zeros((300, 2))], 1).astype(np.float32)
    v, c, n, pi = OP.voxelize_hard(spread, vs, pc_range, max_voxels=10)
    assert len(n) == 10 and pi[:, 0].tolist() == sorted(pi[:, 0].tolist())
