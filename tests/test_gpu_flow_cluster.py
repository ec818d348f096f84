"""GPU parity: fused BEV dynamicness scatter (D1) and z-fit (D3) kernels vs reference fixtures and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import flow_cluster as OF

pytestmark = pytest.mark.gpu


def _g():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "flow_cluster_reference.npz"))


def test_reference_fixtures():
    from liso_amd.kabsch.shape_utils import Shape
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import fit_bev_box_z_and_height_using_points_in_box
    from liso_amd.utils.bev_flow_utils import get_bev_dynamic_flow_map_from_pcl_flow_and_odom

    g = _g()
    c = lambda k: torch.from_numpy(g[k]).cuda()
    dyn, nrf = get_bev_dynamic_flow_map_from_pcl_flow_and_odom(
        pcl_is_valid=c("d1_valid"), pcl=c("d1_pcl"), pillar_coors=c("d1_coors"), point_flow=c("d1_flow"),
        odom_ta_tb=c("d1_odom"), target_shape=(64, 64), return_nonrigid_bev_flow=True)
    assert dyn.shape == (2, 64, 64, 1) and nrf.shape == (2, 64, 64, 3)
    assert np.allclose(dyn.cpu().numpy(), g["d1_dyn"], rtol=1e-4, atol=1e-6)
    assert np.allclose(nrf.cpu().numpy(), g["d1_nrf"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(dyn.cpu().numpy() == 0, g["d1_dyn"] == 0)  # empty pillars are exactly empty
    boxes = Shape(pos=c("d3_pos"), dims=c("d3_dims"), rot=c("d3_rot"), probs=torch.ones(9, 1).cuda())
    num, z, h = fit_bev_box_z_and_height_using_points_in_box(c("d3_pts"), boxes, box_height=1000.0)
    assert np.array_equal(num.cpu().numpy(), g["d3_num"])  # integer: exact
    assert np.allclose(z.cpu().numpy(), g["d3_z"], atol=1e-6) and np.allclose(h.cpu().numpy(), g["d3_h"], atol=1e-6)


@pytest.mark.parametrize("B,N,G", [(1, 120000, 512), (2, 300000, 1024), (1, 7, 64)])
def test_bev_dynamic_flow_vs_oracle_and_determinism(B, N, G):
    from liso_amd.utils.bev_flow_utils import get_bev_dynamic_flow_map_from_pcl_flow_and_odom

    g = torch.Generator().manual_seed(N)
    R = 100.0
    pcl = torch.cat([torch.rand(B, N, 2, generator=g) * R - R / 2, torch.rand(B, N, 2, generator=g)], -1)
    valid = torch.rand(B, N, generator=g) > 0.05
    coors = ((pcl[..., :2] + R / 2) / R * G).to(torch.int32)
    flow = torch.randn(B, N, 3, generator=g)
    odom = torch.eye(4, dtype=torch.float64).repeat(B, 1, 1)
    odom[:, 0, 3] = 1.2
    dyn0, nrf0 = OF.bev_dynamic_flow(valid, pcl, coors, flow, odom, (G, G))
    kw = dict(pcl_is_valid=valid.cuda(), pcl=pcl.cuda(), pillar_coors=coors.cuda(), point_flow=flow.cuda(),
              odom_ta_tb=odom.cuda(), target_shape=(G, G), return_nonrigid_bev_flow=True)
    dyn, nrf = get_bev_dynamic_flow_map_from_pcl_flow_and_odom(**kw)
    assert np.allclose(dyn.cpu().numpy(), dyn0.numpy(), rtol=1e-4, atol=1e-5)
    assert np.allclose(nrf.cpu().numpy(), nrf0.numpy(), rtol=1e-4, atol=1e-5)
    dyn2, nrf2 = get_bev_dynamic_flow_map_from_pcl_flow_and_odom(**kw)
    assert torch.equal(dyn, dyn2) and torch.equal(nrf, nrf2)  # fixed-point atomics: bitwise reproducible
    perm = torch.randperm(N)
    kw2 = dict(kw, pcl_is_valid=valid[:, perm].cuda(), pcl=pcl[:, perm].cuda(), pillar_coors=coors[:, perm].cuda(),
               point_flow=flow[:, perm].cuda())
    dyn3, _ = get_bev_dynamic_flow_map_from_pcl_flow_and_odom(**kw2)
    assert torch.equal(dyn, dyn3)  # and invariant to the order of the points


@pytest.mark.parametrize("N,K", [(120000, 50), (120000, 130), (100, 1), (5000, 0)])
def test_fit_box_z_vs_oracle(N, K):
    from liso_amd.kabsch.shape_utils import Shape
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import fit_bev_box_z_and_height_using_points_in_box

    g = torch.Generator().manual_seed(N + K)
    pts = torch.cat([torch.rand(N, 2, generator=g) * 80 - 40, torch.rand(N, 1, generator=g) * 3 - 2], -1)
    pos = torch.rand(K, 2, generator=g) * 70 - 35
    dims = torch.rand(K, 2, generator=g) * 4 + 1
    rot = (torch.rand(K, 1, generator=g) * 2 - 1) * 3.1
    boxes = Shape(pos=pos.cuda(), dims=dims.cuda(), rot=rot.cuda(), probs=torch.ones(K, 1).cuda())
    num, z, h = fit_bev_box_z_and_height_using_points_in_box(pts.cuda(), boxes)
    if K == 0:
        assert num.numel() == 0
        return
    num0, z0, h0 = OF.fit_box_z(pts, pos, dims, rot[:, 0])
    assert np.array_equal(num.cpu().numpy(), num0.numpy())
    assert np.allclose(z.cpu().numpy(), z0.numpy(), atol=1e-6) and np.allclose(h.cpu().numpy(), h0.numpy(), atol=1e-6)


# ---- D2: DBSCAN over dynamic pillars + region moments, and the assembled FlowClusterDetector -------------------------
def _blobs(seed, G, n_blobs, noise_cells):
    """synthetic dynamic mask + flow: elliptical blobs with a common flow each (some touching / differing only in
    flow), isolated noise cells, a sparse ring that produces border points"""
    g = np.random.default_rng(seed)
    mask = np.zeros((G, G), bool)
    flow = np.zeros((G, G, 3), np.float32)
    rr, cc = np.mgrid[0:G, 0:G]
    for _ in range(n_blobs):
        r0, c0 = g.integers(8, G - 8, 2)
        a, b, th = g.uniform(2, 12), g.uniform(1, 5), g.uniform(-np.pi, np.pi)
        u = (rr - r0) * np.cos(th) + (cc - c0) * np.sin(th)
        v = -(rr - r0) * np.sin(th) + (cc - c0) * np.cos(th)
        m = ((u / a) ** 2 + (v / b) ** 2 <= 1.0) & (g.random((G, G)) < g.uniform(0.35, 1.0))
        mask |= m
        flow[m] = g.normal(0, 0.6, 3).astype(np.float32) + g.normal(0, 0.05, (int(m.sum()), 3)).astype(np.float32)
    nz = g.integers(0, G, (noise_cells, 2))
    mask[nz[:, 0], nz[:, 1]] = True
    flow[nz[:, 0], nz[:, 1]] = g.normal(0, 1.0, (noise_cells, 3)).astype(np.float32)
    return mask, flow


@pytest.mark.parametrize("G,n_blobs,noise,seed", [(128, 12, 60, 0), (512, 60, 400, 1), (64, 3, 5, 2), (96, 0, 30, 3)])
def test_dbscan_labels_match_sklearn_and_regionprops(G, n_blobs, noise, seed):
    """labels bit-identical to sklearn.cluster.DBSCAN called as in flow_cluster_detector.py:151-172 (same numbering,
    same border assignment); region properties equal to the regionprops restatement to fp64 round-off"""
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import cluster_dynamic_pillars, label_region_props
    from liso_amd.utils.bev_utils import get_metric_voxel_center_coords
    from oracle.flow_cluster import dbscan_bev_labels, regionprops_restated

    R = G * 100.0 / 512.0
    centers = get_metric_voxel_center_coords(np.float32(R), np.float32(R), np.array([G, G], np.int32)).astype(np.float32)
    masks, flows = zip(*[_blobs(seed * 10 + b, G, n_blobs, noise) for b in range(2)])
    mask_t, flow_t = torch.from_numpy(np.stack(masks)).cuda(), torch.from_numpy(np.stack(flows)).cuda()
    ct = torch.from_numpy(centers).cuda()
    labels, num = cluster_dynamic_pillars(mask_t, flow_t, ct[:, 0, 0], ct[0, :, 1])
    labels2, _ = cluster_dynamic_pillars(mask_t, flow_t, ct[:, 0, 0], ct[0, :, 1])
    assert torch.equal(labels, labels2)  # lock-free union-find, deterministic result
    kmax = max(int(num.max()), 1)
    props = label_region_props(labels, kmax).cpu().numpy()
    for b in range(2):
        ref = dbscan_bev_labels(masks[b], flows[b], centers[..., :2])
        assert int(num[b]) == int(ref.max())
        assert np.array_equal(labels[b].cpu().numpy(), ref)
        rp = regionprops_restated(ref)
        assert rp.shape[0] == int(num[b])
        if rp.shape[0]:
            np.testing.assert_allclose(props[b, :rp.shape[0]], rp, rtol=1e-9, atol=1e-9)


def _chains(seed, G, n_chains):
    """thin, branching, zig-zag components whose cells' roots are NOT ordered along the walk (anti-diagonals, combs, random walks that
    turn back): the shapes on which a union that re-parents a non-root can cut a tree off"""
    g = np.random.default_rng(seed)
    mask = np.zeros((G, G), bool)
    for _ in range(n_chains):
        r, c = g.integers(4, G - 4, 2)
        kind = g.integers(0, 3)
        for step in range(int(g.integers(10, 60))):
            mask[r, c] = True
            if kind == 0:  # anti-diagonal staircase: row index falls while the column rises
                r, c = r - (step & 1), c + 1
            elif kind == 1:  # comb: a spine with teeth pointing to SMALLER rows
                mask[max(r - 2, 0):r, c] = step % 2 == 0
                c += 1
            else:  # random walk with steps of up to 2 cells (gaps that only the window bridges)
                r, c = r + int(g.integers(-2, 3)), c + int(g.integers(-2, 3))
            r, c = int(np.clip(r, 0, G - 1)), int(np.clip(c, 0, G - 1))
    flow = np.zeros((G, G, 3), np.float32)
    flow[mask] = g.normal(0, 0.02, (int(mask.sum()), 3)).astype(np.float32)
    return mask, flow


@pytest.mark.parametrize("G,pitch,min_samples,seed", [(96, 0.6, 2, 0), (128, 0.45, 3, 1), (200, 0.3, 3, 2), (64, 0.95, 2, 3)])
def test_dbscan_chain_shaped_clusters_match_sklearn(G, pitch, min_samples, seed):
    """the LDS union-find of dbscan_union_tiled_kernel on thin chains with small windows (eps / pitch = 1..3 cells): a failed hook has
    to go on with the displaced parent, or a sub-tree is silently cut off (round-5 advisor finding; the blob fixtures never hit it)"""
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import cluster_dynamic_pillars
    from oracle.flow_cluster import dbscan_bev_labels

    ax = ((np.arange(G) - G / 2 + 0.5) * pitch).astype(np.float32)
    centers = np.stack(np.meshgrid(ax, ax, indexing="ij"), -1)
    masks, flows = zip(*[_chains(seed * 7 + b, G, 40) for b in range(3)])
    mask_t, flow_t = torch.from_numpy(np.stack(masks)).cuda(), torch.from_numpy(np.stack(flows)).cuda()
    xs, ys = torch.from_numpy(ax).cuda(), torch.from_numpy(ax).cuda()
    refs = [dbscan_bev_labels(masks[b], flows[b], centers, eps=1.0, min_samples=min_samples) for b in range(3)]
    for _ in range(5):  # (the lost union was a race: several runs)
        labels, num = cluster_dynamic_pillars(mask_t, flow_t, xs, ys, eps=1.0, min_samples=min_samples, pitch=pitch)
        for b in range(3):
            assert int(num[b]) == int(refs[b].max())
            assert np.array_equal(labels[b].cpu().numpy(), refs[b])


def test_flow_cluster_detector_matches_oracle_and_finds_movers():
    """FlowClusterDetector.forward (flow_cluster_detector.py:87-336) end to end on two synthetic 120k-point sweeps:
    same boxes as the CPU restatement (positions/dims exact to fp32 round-off, heading/velocity <= 1e-3), and the
    mined boxes sit on objects that really move"""
    from liso_amd.datasets.synthetic import cluster_sample
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector
    from liso_amd.utils.config import default_cfg
    from oracle.flow_cluster import flow_cluster_detector_forward

    cfg = default_cfg(grid=512, bev_range_m=100.0)
    det = FlowClusterDetector(cfg).cuda()
    sample, scenes = cluster_sample(5, torch.device("cuda"), batch=2, n_points=120000)
    boxes = det(sample, global_step=1)
    cpu = lambda t: t.detach().cpu()
    ref = flow_cluster_detector_forward(
        cpu(sample["pcl_ta"]["pcl"]), cpu(sample["pcl_ta"]["pcl_is_valid"]), cpu(sample["pcl_full_w_ground_ta"]),
        cpu(sample["pcl_ta"]["pillar_coors"]), cpu(sample["gt"]["flow_ta_tb"]), cpu(sample["gt"]["odom_ta_tb"]),
        cpu(sample["src_trgt_time_delta_s"]), det.pcl_bev_center_coords_homog_np[..., :2], det.bev_pixel_per_meter_res_np)
    assert np.array_equal(det.last_bev_labels.cpu().numpy(), ref["labels"])
    assert boxes.valid.shape == ref["valid"].shape and torch.equal(cpu(boxes.valid), ref["valid"])
    assert int(boxes.valid.sum()) >= 4
    v = ref["valid"]
    assert torch.allclose(cpu(boxes.pos)[v], ref["pos"][v], atol=1e-4)
    assert torch.allclose(cpu(boxes.dims)[v].double(), ref["dims"][v], atol=1e-4)
    assert torch.allclose(cpu(boxes.velo)[v].double(), ref["velo"][v], atol=2e-3)
    dth = (cpu(boxes.rot)[v].double() - ref["rot"][v] + np.pi) % (2 * np.pi) - np.pi
    assert float(dth.abs().max()) < 2e-3
    # every mined box lies on a moving scene object (within its footprint diagonal), moving at about its speed
    for b, (sb, speed) in enumerate(scenes):
        for k in torch.nonzero(boxes.valid[b])[:, 0]:
            d = (sb[:, :2] - boxes.pos[b, k, :2].float()).norm(dim=-1)
            j = int(d.argmin())
            assert float(d[j]) < 3.5 and float(speed[j]) > 0.1, (b, int(k), float(d[j]), float(speed[j]))


def test_odom_inverse_minus_eye_kernel_matches_lu_inverse():
    """liso_odom_inverse_minus_eye_f64 (cofactor expansion, one launch, capturable) vs torch.linalg.inv(odom) - I in fp64
    (bev_flow_utils.py:30-33): rigid transforms as the datasets provide them and general well-conditioned 4x4 matrices"""
    import math

    from liso_amd.utils.bev_flow_utils import odometry_minus_identity

    g = torch.Generator().manual_seed(3)
    mats = []
    for _ in range(33):
        th, tx, ty, tz = [float(v) for v in (torch.rand(4, generator=g, dtype=torch.float64) - 0.5) * torch.tensor([6.28, 40.0, 40.0, 2.0], dtype=torch.float64)]
        T = torch.eye(4, dtype=torch.float64)
        T[0, 0], T[0, 1], T[1, 0], T[1, 1] = math.cos(th), -math.sin(th), math.sin(th), math.cos(th)
        T[0, 3], T[1, 3], T[2, 3] = tx, ty, tz
        mats.append(T)
    for _ in range(31):
        mats.append(torch.eye(4, dtype=torch.float64) * 2.0 + torch.randn(4, 4, generator=g, dtype=torch.float64) * 0.4)
    m = torch.stack(mats)
    got = odometry_minus_identity(m.cuda()).cpu()
    want = torch.linalg.inv(m) - torch.eye(4, dtype=torch.float64)
    assert got.dtype == torch.float64 and got.shape == (64, 4, 4)
    assert float((got - want).abs().max()) <= 1e-12 * float(want.abs().max())
    # fp32 odometry (some loaders) is promoted like the reference's .double()
    got32 = odometry_minus_identity(m[:4].float().cuda()).cpu()
    assert float((got32 - (torch.linalg.inv(m[:4].float().double()) - torch.eye(4, dtype=torch.float64))).abs().max()) <= 1e-12 * 50
