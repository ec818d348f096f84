"""Local box refinement on the device (include/liso_tracking.h: liso_fit_boxes_closeness_f32; liso_amd/tracker/tracking.py) against
the fixture written by the reference's own functions and against the CPU oracle at sweep size."""
import numpy as np
import pytest
import torch

from tests.test_oracle_box_refinement import load_track

pytestmark = pytest.mark.gpu


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _cfg(d):
    return _Cfg({k: _cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def test_rectangle_fit_matches_reference_fixture(golden_dir):
    from liso_amd.tracker.tracking import fit_boxes_to_points

    g = np.load(f"{golden_dir}/box_refinement_reference.npz")
    for i in range(5):
        pts = torch.from_numpy(g[f"fit{i}_points"]).float().cuda()
        # a box that contains every point: the fit then sees exactly the fixture's cluster (in float32 coordinates)
        lo, hi = pts[:, :2].min(0).values, pts[:, :2].max(0).values
        box = torch.tensor([[float((lo[0] + hi[0]) / 2), float((lo[1] + hi[1]) / 2), 0.0, 200.0, 200.0, 2.0, 0.0]], device="cuda")
        count, fit = fit_boxes_to_points(pts, box, 1.0)
        assert int(count[0]) == pts.shape[0]
        from oracle import box_refinement as ob
        center, length, width, yaw = ob.closeness_fit(pts[:, :2].double().cpu().numpy())
        assert np.allclose(fit[0].cpu().numpy(), [center[0], center[1], length, width, yaw], rtol=0, atol=1e-9), i
        assert np.allclose(fit[0].cpu().numpy(), g[f"fit{i}_result"], rtol=0, atol=2e-5), i  # (fixture points are float64)


@pytest.mark.parametrize("tag", ["t0", "t1", "t2"])
def test_track_refinement_matches_reference_fixture(golden_dir, tag):
    from liso_amd.kabsch.shape_utils import Shape
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector
    from liso_amd.tracker.tracking import perform_local_box_refinement

    g = np.load(f"{golden_dir}/box_refinement_reference.npz")
    age, start, fit_rot, fit_pos, q, clouds = load_track(g, tag)
    cfg = _cfg({"data": {"tracking_cfg": {"fit_box_to_points": {"fit_rot": fit_rot, "fit_pos": fit_pos, "fitting_dims_bloat_factor": 1.2}}}})
    boxes = Shape(pos=torch.from_numpy(g[f"{tag}_in_pos"]).cuda(), dims=torch.from_numpy(g[f"{tag}_in_dims"]).cuda(),
                  rot=torch.from_numpy(g[f"{tag}_in_rot"]).cuda(), probs=torch.ones(age, 1, device="cuda"))
    predictor = FlowClusterDetector.__new__(FlowClusterDetector) if q == 0.95 else object()
    res = perform_local_box_refinement(cfg, predictor, [torch.from_numpy(c).cuda() for c in clouds], boxes, age, start)
    assert res is boxes  # in place, like the reference
    assert np.allclose(res.rot.cpu().numpy(), g[f"{tag}_out_rot"], rtol=0, atol=1e-6)
    assert np.allclose(res.dims.cpu().numpy(), g[f"{tag}_out_dims"], rtol=0, atol=1e-6)
    assert np.allclose(res.pos.cpu().numpy(), g[f"{tag}_out_pos"], rtol=0, atol=5e-6)


def test_all_boxes_of_a_frame_at_sweep_size_equal_the_oracle():
    """120k-point sweep, 60 boxes in one launch (dense cars, empty boxes, a box holding a wall of 20k points, NaN padding rows)"""
    from liso_amd.tracker.tracking import fit_boxes_to_points
    from oracle import box_refinement as ob

    g = np.random.default_rng(8)
    N, K = 120000, 60
    pos = np.concatenate([g.uniform(-40, 40, (K, 2)), np.full((K, 1), -0.8)], -1)
    dims = np.stack([g.uniform(3.5, 5.5, K), g.uniform(1.6, 2.2, K), g.uniform(1.4, 1.9, K)], -1)
    yaw = g.uniform(-np.pi, np.pi, K)
    cloud = np.concatenate([g.uniform(-50, 50, (N, 2)), g.uniform(-2, 1, (N, 1)), g.uniform(0, 1, (N, 1))], -1)
    for k in range(0, K, 2):  # every other box holds an L-shaped cluster
        n = int(g.integers(20, 600))
        t = g.uniform(-0.5, 0.5, n)
        side = g.uniform(size=n) < 0.6
        x = np.where(side, t * dims[k, 0], 0.5 * dims[k, 0]) + g.normal(0, 0.03, n)
        y = np.where(side, -0.5 * dims[k, 1], t * dims[k, 1]) + g.normal(0, 0.03, n)
        c, s = np.cos(yaw[k]), np.sin(yaw[k])
        sel = g.choice(N, n, replace=False)
        cloud[sel, 0], cloud[sel, 1] = pos[k, 0] + c * x - s * y, pos[k, 1] + s * x + c * y
    dims[1, :2] = [60.0, 30.0]  # a box that swallows ~20k points
    cloud[-5:, :2] = np.nan
    cloud = cloud.astype(np.float32)
    boxes7 = np.concatenate([pos, dims, yaw[:, None]], -1).astype(np.float32)
    count, fit = fit_boxes_to_points(torch.from_numpy(cloud).cuda(), torch.from_numpy(boxes7).cuda(), 1.2)
    count, fit = count.cpu().numpy(), fit.cpu().numpy()
    assert count[1] > 10000
    for k in range(K):
        inside = ob.points_in_bloated_footprint(cloud[:, :3], boxes7[k, :3], boxes7[k, 3:6], boxes7[k, 6], 1.2)
        assert abs(int(inside.sum()) - int(count[k])) <= 1, k  # (a point within one fp64 ulp of the footprint edge may flip)
        if inside.sum() == 0:
            assert np.isnan(fit[k]).all()
        elif int(inside.sum()) == int(count[k]):
            center, length, width, ang = ob.closeness_fit(cloud[inside, :2].astype(np.float64))
            assert np.allclose(fit[k], [center[0], center[1], length, width, ang], rtol=0, atol=1e-8), k
