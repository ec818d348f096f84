"""Target rendering on the device (include/liso_detector.h: liso_render_center_targets_f32) against the reference fixture
and the torch formulation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hip_target_maps_match_reference_fixture(golden_dir):
    from liso_amd.datasets.targets import render_center_targets

    g = np.load(f"{golden_dir}/targets_reference.npz")
    for tag in ("a", "b", "c"):
        G, R = int(g[f"{tag}_grid_range"][0]), float(g[f"{tag}_grid_range"][1])
        pos, dims, rot = (torch.from_numpy(g[f"{tag}_box_{k}"]).float()[None].cuda() for k in ("pos", "dims", "rot"))
        valid = torch.ones(pos.shape[:2], dtype=torch.bool, device="cuda")
        maps = render_center_targets(pos, dims, rot, valid, (G, G), (R, R))
        for k in ("probs", "dims", "pos", "rot"):
            want, got = g[f"{tag}_{k}"], maps[k][0].cpu().numpy()
            assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0), (tag, k, np.abs(got - want).max())
        assert np.array_equal(maps["center_bool_mask"][0].cpu().numpy(), g[f"{tag}_center_bool_mask"]), tag


@pytest.mark.parametrize("B,K,G", [(4, 15, 128), (2, 100, 128), (1, 0, 64), (3, 7, 256)])
def test_hip_target_maps_equal_torch_formulation(B, K, G):
    from liso_amd.datasets.targets import render_center_targets, render_center_targets_torch

    g = torch.Generator().manual_seed(B * 100 + K)
    R = 100.0
    pos = torch.cat([torch.rand(B, K, 2, generator=g) * 0.9 * R - 0.45 * R, torch.rand(B, K, 1, generator=g) - 1.5], -1).cuda()
    dims = torch.stack([torch.rand(B, K, generator=g) * 3 + 2.5, torch.rand(B, K, generator=g) + 1.4, torch.rand(B, K, generator=g) + 1.2], -1).cuda()
    rot = ((torch.rand(B, K, 1, generator=g) * 2 - 1) * 3.1).cuda()
    valid = (torch.rand(B, K, generator=g) > 0.25).cuda()
    a = render_center_targets(pos, dims, rot, valid, (G, G), (R, R))
    if K == 0:  # no box slot at all: all-background maps (the torch formulation cannot reduce over an empty axis)
        assert all(float(a[k].abs().sum()) == 0 for k in ("probs", "dims", "pos", "rot")) and not bool(a["center_bool_mask"].any())
        return
    b = render_center_targets_torch(pos, dims, rot, valid, (G, G), (R, R))
    for k in ("probs", "dims", "pos", "rot"):
        assert a[k].shape == b[k].shape
        # cells where two boxes tie for the maximum to the last bit may resolve differently: none expected, allow a few
        bad = ((a[k] - b[k]).abs() > 1e-4 * max(float(b[k].abs().max()), 1.0)).any(dim=-1)
        assert int(bad.sum()) <= 2, (k, int(bad.sum()))
    assert torch.equal(a["center_bool_mask"], b["center_bool_mask"])
