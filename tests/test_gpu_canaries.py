"""Memory-safety evidence without GPU AddressSanitizer (round-4 VERDICT item 8): the device ops run with every wrapper-allocated
output and workspace between canary guard bands (tests/guarded_alloc.py), at nominal sizes and at the capacity limits the kernels
promise to respect -- more occupied pillars than max_voxels, more clusters than box slots, more occupied cells than the sparse
convolution's cell lists hold, crowded pillars, empty clouds, NaN rows -- and the guards must come back untouched.
The harness itself is checked first: a deliberate out-of-bounds write must be reported."""
import numpy as np
import pytest
import torch

from tests.guarded_alloc import GUARD, guarded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_harness_reports_an_out_of_bounds_write():
    with pytest.raises(AssertionError, match="guard overwritten"):
        with guarded():
            t = torch.empty(1000, dtype=torch.float32, device=DEV)
            # one float behind the payload, through the storage (what a kernel with a wrong bound does)
            raw = torch.empty(0, dtype=torch.float32, device=DEV).set_(t.untyped_storage(), t.storage_offset() + 1000, (1,))
            raw.fill_(1.0)
    with guarded() as g:  # and stays silent for in-bounds work
        t = torch.zeros((7, 13), dtype=torch.float32, device=DEV)
        t += 1.0
        assert g.check() == 1 and float(t.sum()) == 91.0


def _pfn(grid, rng):
    from liso_amd.networks.pcl_to_feature_grid.pcl_to_feature_grid import PointsPillarFeatureNetWrapper
    from liso_amd.utils.config import default_cfg

    cfg = default_cfg(grid=grid, bev_range_m=rng)
    torch.manual_seed(0)
    return cfg, PointsPillarFeatureNetWrapper(cfg).to(DEV).train()


def test_voxeliser_and_pfn_beyond_the_pillar_cap_crowded_pillars_nan_rows_and_empty_clouds():
    cfg, net = _pfn(512, 100.0)
    g = torch.Generator().manual_seed(3)
    uniform = torch.rand(200000, 4, generator=g) * torch.tensor([100.0, 100, 3, 1]) - torch.tensor([50.0, 50, 1.5, 0])  # ~140k occupied pillars
    crowded = torch.cat([torch.rand(60000, 4, generator=g) * torch.tensor([0.1, 0.1, 1, 1]),       # 60k points in ONE pillar
                         torch.rand(20000, 4, generator=g) * torch.tensor([100.0, 100, 3, 1]) - torch.tensor([50.0, 50, 1.5, 0])])
    nans = uniform[:50000].clone()
    nans[::3] = float("nan")
    clouds = [uniform, crowded, torch.zeros(0, 4), nans]
    with guarded() as gd:
        bev, occ = net([c.to(DEV) for c in clouds])
        gd.check()
        assert torch.isfinite(bev).all() and float(occ[2].sum()) == 0
        n_pillars = occ.flatten(1).sum(1)
        assert float(n_pillars[0]) == 40000.0, "the first-come cap of 40 000 pillars"  # (voxel_generator.py:266-270)
        bev.float().square().sum().backward()
        gd.check()


def test_sparse_canvas_convolution_beyond_its_cell_capacity():
    from liso_amd.utils import mfma_conv as MC

    B, H = 1, 512
    x = torch.zeros(B, 64, H, H, device=DEV).contiguous(memory_format=torch.channels_last)
    occ = (torch.rand(B, 1, H, H, device=DEV) < 0.3)  # 78 000 occupied cells > SPARSE_STEM_MAX_CELLS = 40 960
    x = torch.where(occ, torch.randn_like(x), x)
    conv = torch.nn.Conv2d(64, 32, 7, stride=2, padding=3).to(DEV)
    assert int(occ.sum()) > MC.SPARSE_STEM_MAX_CELLS
    try:
        with guarded() as gd:
            y = MC.conv2d(conv, x, relu=True, occupancy=occ.float())
            gd.check()
        assert torch.isfinite(y).all()
        assert MC.sparse_stem_overflowed(torch.device(DEV)), "more occupied cells than the lists hold must raise the overflow flag"
    finally:
        MC.reset_sparse_stem_overflow(torch.device(DEV))  # (the flag is sticky per process: later tests assert that it is clear)


def test_more_clusters_than_box_slots_and_scattered_labels():
    from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import cluster_dynamic_pillars, label_region_props
    from liso_amd.networks.flow_cluster_detector import mining_ops as MO
    from liso_amd.utils.bev_utils import get_metric_voxel_center_coords

    G, K = 256, 64
    rng = np.random.default_rng(5)
    mask = np.zeros((1, G, G), bool)
    flow = np.zeros((1, G, G, 3), np.float32)
    for k in range(150):  # 150 separate movers: more clusters than the 64 fixed slots
        r0, c0 = rng.integers(4, G - 8, 2)
        mask[0, r0:r0 + 4, c0:c0 + 4] = True
        flow[0, r0:r0 + 4, c0:c0 + 4, 0] = rng.uniform(-2, 2)
    mask[0][rng.random((G, G)) < 0.05] = True  # + scattered single pillars
    centers = get_metric_voxel_center_coords(np.float32(100.0), np.float32(100.0), np.array([G, G], np.int32)).astype(np.float32)
    ct = torch.from_numpy(centers).to(DEV)
    with guarded() as gd:
        labels, num = cluster_dynamic_pillars(torch.from_numpy(mask).to(DEV), torch.from_numpy(flow).to(DEV), ct[:, 0, 0].contiguous(),
                                              ct[0, :, 1].contiguous())
        gd.check()
        assert int(num.max()) > K
        props = label_region_props(labels, K)  # labels beyond K are dropped, never written
        gd.check()
        center, dims2, rot1, dims2_f32, rot1_f32 = MO.boxes_from_regions(props, ct[:, 0, 0].contiguous(), ct[0, :, 1].contiguous(),
                                                                         np.array([G / 100.0, G / 100.0], np.float32))
        gd.check()
    assert props.shape == (1, K, 5) and torch.isfinite(props).all()


def test_fused_loop_step_eager_under_guards():
    """the whole fused iteration, launched eagerly (SLIM inference, box mining with 64 fixed slots, target maps, detector step)"""
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.trainer import LisoLoopTrainer
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

    dev = torch.device(DEV)
    torch.manual_seed(0)
    tr = LisoLoopTrainer(apply_slim_simple_knn_training(default_cfg(grid=256, bev_range_m=50.0)), dev, compute_dtype=torch.bfloat16,
                         total_steps=10, use_graph=False, overlap=False)
    pairs = [slim_pair(60 + i, dev, n_points=30000, grid=256, bev_range_m=50.0) for i in range(2)]
    float(tr.step(*pairs[0]))  # lazy initialisations outside the guards
    with guarded() as gd:
        loss = float(tr.step(*pairs[1]))
        n = gd.check()
    assert np.isfinite(loss) and n > 50


def test_slim_train_step_eager_under_guards():
    """pillars -> RAFT (6 iterations) -> decoder -> exact 1-NN loss -> backward -> RMSprop: kNN buckets, split-K slabs, correlation"""
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.trainer import SlimTrainer
    from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg

    dev = torch.device(DEV)
    torch.manual_seed(0)
    st = SlimTrainer(apply_slim_simple_knn_training(default_cfg(grid=128, bev_range_m=40.0)), dev, use_graph=False)
    s0, s1 = slim_pair(9, dev, n_points=12000, grid=128, bev_range_m=40.0)
    float(st.step(s0, s1))
    with guarded() as gd:
        loss = float(st.step(s0, s1))
        n = gd.check()
    assert np.isfinite(loss) and n > 100


def test_iou3d_nms_buffers():
    from liso_amd import iou3d_nms_cuda as M
    from oracle import iou3d as O

    b, s = O.random_boxes(3000, 1, 30.0)
    tb = torch.from_numpy(b[np.argsort(-s, kind="stable")]).to(DEV)
    with guarded() as gd:
        iou = torch.zeros(3000, 3000, device=DEV)
        M.boxes_iou_bev_gpu(tb, tb, iou)
        keep = torch.zeros(3000, dtype=torch.int64)
        n = M.nms_gpu(tb, keep, 0.1)
        gd.check()
    assert 0 < n <= 3000


def test_voxel_slots_at_the_largest_slot_count_and_crowded_pillars():
    """max_points = 32, the largest the ABI takes (round-5 advisor finding on voxel_slots_kernel: on its crowded path, n > 64 points, the
    rounds ran to max_points whatever n -- harmless while max_points <= 32 < 64 is enforced, bounded by n since round 6; a larger
    max_points is refused).  Pillars of 3 .. 500 points against a host construction: the first min(n, 32) slots are the pillar's point
    indices in ascending order, the slots behind them keep what the caller put there (-7)."""
    import ctypes

    import liso_amd.networks.pcl_to_feature_grid.pcl_to_feature_grid as P
    from liso_amd import _lib as L

    _, net = _pfn(64, 64.0)
    net.max_num_points = 32
    pcfg = net._pcfg(4)
    g = torch.Generator().manual_seed(5)
    sizes = [3, 31, 32, 33, 64, 65, 500]
    pts = []
    for k, n in enumerate(sizes):  # pillar k: cell (2 k + 1, 5), 1-m cells on [-32, 32)^2
        pts.append(torch.rand(n, 4, generator=g) * torch.tensor([0.9, 0.9, 2.0, 1.0]) + torch.tensor([-32.0 + 2 * k + 1, -32.0 + 5, -1.0, 0.0]))
    cloud = torch.cat(pts)[torch.randperm(sum(sizes), generator=g)].to(DEV).contiguous()
    with guarded() as gd:
        real_empty = torch.empty

        def poisoned(*a, **k):  # the slots buffer starts as -7: untouched entries stay recognisable
            t = real_empty(*a, **k)
            if k.get("dtype") == torch.int32 and t.dim() == 2 and t.shape[1] == 32:
                t.fill_(-7)
            return t

        P.torch.empty = poisoned
        try:
            coors, num_points, slots, num_voxels, _ = P.voxelize_raw(cloud, [0, cloud.shape[0]], pcfg)
        finally:
            P.torch.empty = real_empty
        gd.check()
    nv = int(num_voxels[0])
    assert nv == len(sizes)
    cell = ((cloud[:, 0] + 32).floor().long() * 64 + (cloud[:, 1] + 32).floor().long()).cpu()
    for v in range(nv):
        idx = torch.nonzero(cell == int(coors[v, 2]) * 64 + int(coors[v, 3])).flatten()
        kept = min(idx.numel(), 32)
        assert int(num_points[v]) == kept, (v, idx.numel(), int(num_points[v]))
        got = slots[v].cpu()
        assert torch.equal(got[:kept].long(), idx[:kept]), (v, idx.numel())
        assert bool((got[kept:] == -7).all()), (v, idx.numel(), got[kept:kept + 5])
    pcfg.max_points = 33  # beyond the one-wave-per-pillar mapping: refused, not truncated
    off = (ctypes.c_int * 2)(0, cloud.shape[0])
    ws = torch.empty(1 << 22, dtype=torch.uint8, device=DEV)
    rc = L.lib().liso_pillars_voxelize_f32(L.ptr(cloud), off, 1, ctypes.byref(pcfg), L.ptr(coors), L.ptr(num_points), L.ptr(slots),
                                           L.ptr(num_voxels), L.ptr(ws), L.ptr(ws), ws.numel(), L.stream_ptr())
    assert rc != 0
