"""Box-snippet augmentation on the device (include/liso_augment.h, liso_amd/datasets/box_augmentation.py) against the fixture
generated from the reference's `create_augmented_sample_from_box_snippet_db` and against the CPU oracle at BASELINE's sizes."""
import numpy as np
import pytest
import torch

from tests.test_oracle_box_augment import load_case

pytestmark = pytest.mark.gpu


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def _cfg(d):
    return _Cfg({k: _cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def make_cfg(G, R, box_cfg, flow_source="slim_flow", network="centerpoint"):
    return _cfg({"data": {"bev_range_m": [R, R], "img_grid_size": [G, G], "flow_source": flow_source, "train_on_box_source": "mined",
                          "limit_pillar_height": False, "augmentation": {"boxes": dict(box_cfg, active=True)}},
                 "network": {"name": network}, "loss": {"supervised": {"centermaps": {"confidence_target": "gaussian"}}}})


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_augmented_sample_matches_the_reference_fixture(golden_dir, tag):
    from liso_amd.datasets.box_augmentation import BoxAugmenter, BoxSnippetDb
    from liso_amd.kabsch.shape_utils import Shape

    g = np.load(f"{golden_dir}/box_augment_reference.npz")
    G, R, db, box_cfg, need_flow, seed, n_pre = load_case(g, tag)
    flow_source = str(g[f"{tag}_flow_source"])
    boxes = Shape(**{k: torch.from_numpy(g[f"{tag}_db_box_{k}"]) for k in ("pos", "dims", "rot", "probs")})
    rows = g[f"{tag}_db_rows"]
    off = np.concatenate([[0], np.cumsum(g[f"{tag}_db_counts"])])
    sdb = BoxSnippetDb({"pcl_in_box_cosy": db["points"], "boxes": boxes, "lidar_rows": [rows[off[i]:off[i + 1]] for i in range(len(off) - 1)],
                        "box_T_sensor": g[f"{tag}_db_box_T_sensor"]}, "cuda")
    aug = BoxAugmenter(make_cfg(G, R, box_cfg, flow_source), sdb, need_flow=need_flow)
    pcl = torch.from_numpy(g[f"{tag}_in_pcl"]).cuda()
    sample = {"pcl_ta": {"pcl": pcl, "pillar_coors": torch.from_numpy(g[f"{tag}_in_coors"]).cuda()},
              "pcl_full_w_ground_ta": torch.cat([pcl, pcl[:50]]), "pcl_full_no_ground_ta": pcl.clone(),
              "gt": {"odom_ta_tb": torch.eye(4, dtype=torch.float64)},
              flow_source: {"flow_ta_tb": torch.from_numpy(g[f"{tag}_in_flow"]).cuda()}}
    pre = None
    if n_pre:
        pre = Shape(**{k: torch.from_numpy(g[f"{tag}_pre_{k}"]) for k in ("pos", "dims", "rot", "probs", "velo")})
    np.random.seed(seed)
    torch.manual_seed(seed)
    res = aug.create_augmented_sample_from_box_snippet_db(0.1, sample, prediscovered_boxes=pre)

    # integer / index outputs: identical
    assert np.array_equal(res["pcl_ta"]["pillar_coors"].cpu().numpy(), g[f"{tag}_out_coors"])
    b = res["gt"]["boxes"]
    assert np.array_equal(np.asarray(b.valid), g[f"{tag}_out_box_valid"])
    for k in ("pos", "dims", "rot", "probs"):  # host arithmetic on the reference's draws: identical
        assert np.array_equal(getattr(b, k).numpy(), g[f"{tag}_out_box_{k}"]), k
    # float outputs of the kernels: float64 arithmetic rounded once to float32 -> identical up to a rounding tie (<= 1 ulp)
    def close(got, want, what):
        got = got.cpu().numpy()
        assert got.shape == want.shape, (what, got.shape, want.shape)
        assert np.all(np.abs(got - want) <= np.spacing(np.abs(want).astype(np.float32))), (what, np.abs(got - want).max())
        assert np.mean(got == want) > 0.999, (what, np.mean(got == want))
    close(res["pcl_ta"]["pcl"], g[f"{tag}_out_pcl"], "pcl")
    close(res["pcl_full_w_ground_ta"], g[f"{tag}_out_pcl_full_w_ground"], "full_w_ground")
    close(res["pcl_full_no_ground_ta"], g[f"{tag}_out_pcl_full_no_ground"], "full_no_ground")
    close(b.velo, g[f"{tag}_out_box_velo"], "velo")
    if need_flow:
        close(res[flow_source]["flow_ta_tb"], g[f"{tag}_out_flow"], "flow")
    mined = res["mined"]
    assert mined["prediscovered_boxes"].pos.shape[0] == int(g[f"{tag}_out_n_prediscovered"])
    for k in ("probs", "dims", "pos", "rot"):
        want, got = g[f"{tag}_out_centermaps_{k}"], mined[f"centermaps_{k}"].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0), (k, np.abs(got - want).max())
    assert np.array_equal(mined["centermaps_center_bool_mask"].cpu().numpy(), g[f"{tag}_out_centermaps_center_bool_mask"])


@pytest.mark.parametrize("G,R,n,radius", [(512, 100.0, 120000, 10), (1024, 100.0, 300000, 20), (64, 40.0, 0, 3), (96, 30.0, 500, 40),
                                          (130, 50.0, 2000, 5)])
def test_free_mask_and_selection_equal_the_oracle(G, R, n, radius):
    """BASELINE's grids (512^2 / 120k points, 1024^2 / 300k points), an empty sweep, a radius wider than a bitmap word and a
    grid width that is not a multiple of 64: mask, per-row prefix and the k-th-free-cell selection are exact"""
    from liso_amd.datasets.box_augmentation import free_location_mask, select_free_cells
    from oracle import box_augment as ob

    rs = np.random.default_rng(G + n)
    W = G if G != 130 else 70
    coors = np.stack([rs.integers(0, G, n), rs.integers(0, W, n)], -1).astype(np.int32)
    if n:
        coors[: n // 2] = (coors[: n // 2] * np.array([0.3, 0.3])).astype(np.int32)  # a dense corner and a sparse rest
        coors[-1] = [-1, 5]  # a row outside the grid is ignored
    free, prefix = free_location_mask(torch.from_numpy(coors).cuda(), (G, W), radius)
    want = ob.free_location_mask(coors[:-1] if n else coors, (G, W), radius)
    assert np.array_equal(free.cpu().numpy().astype(bool), want)
    assert np.array_equal(prefix.cpu().numpy(), np.concatenate([[0], np.cumsum(want.sum(1))]))
    nfree = int(want.sum())
    if nfree:
        idx = np.unique(np.concatenate([[0, nfree - 1], rs.integers(0, nfree, 40)]))
        got = select_free_cells(free, prefix, idx).cpu().numpy()
        assert np.array_equal(got, np.flatnonzero(want.reshape(-1))[idx])
        assert select_free_cells(free, prefix, np.array([nfree, -1])).cpu().tolist() == [-1, -1]


def test_paste_at_database_scale_equals_the_oracle_arithmetic():
    """15 objects of up to 2000 points out of a 3000-snippet database: gather, pose, flow and speed against float64 numpy"""
    from liso_amd.datasets.box_augmentation import BoxSnippetDb, paste_snippets
    from liso_amd.kabsch.shape_utils import Shape

    rs = np.random.default_rng(5)
    M = 3000
    counts = rs.integers(11, 2000, M)
    pcls = [rs.normal(size=(c, 4)).astype(np.float32) for c in counts]
    boxes = Shape(pos=torch.zeros(M, 3), dims=torch.ones(M, 3), rot=torch.zeros(M, 1), probs=torch.ones(M, 1))
    db = BoxSnippetDb({"pcl_in_box_cosy": pcls, "boxes": boxes}, "cuda")
    objs = rs.integers(0, M, 15)
    sel = [rs.permutation(counts[o])[: max(1, counts[o] // 2)] for o in objs]
    offs = np.concatenate([[0], np.cumsum([len(s) for s in sel])])
    pose = rs.normal(size=(15, 3, 4)) * 10
    rnd = rs.random((offs[-1], 3))
    pts, flow, velo = paste_snippets(db, np.concatenate([db.offsets[o] + s for o, s in zip(objs, sel)]), offs, pose, rnd, 1.0, 3.0)
    for i, (o, s) in enumerate(zip(objs, sel)):
        p = pcls[o][s]
        want = np.concatenate([np.einsum("ij,nj->ni", pose[i], np.concatenate([p[:, :3], np.ones_like(p[:, :1])], -1).astype(np.float64)),
                               p[:, 3:]], -1).astype(np.float32)
        got = pts[offs[i]:offs[i + 1]].cpu().numpy()
        assert np.all(np.abs(got - want) <= np.spacing(np.abs(want))), i
        fl = 1.0 + rnd[offs[i]:offs[i + 1]] * 2.0
        assert np.array_equal(flow[offs[i]:offs[i + 1]].cpu().numpy(), fl.astype(np.float32))
        assert abs(float(velo[i]) - np.linalg.norm(fl, axis=-1).mean()) <= 2e-7 * 3.0


def test_raydrop_variant_consumes_the_generator_like_the_reference():
    """`use_raydrop_augm`: rows are kept by LiDAR layer; a seeded run is reproducible and pastes a subset of each snippet"""
    from liso_amd.datasets.box_augmentation import BoxAugmenter, BoxSnippetDb
    from liso_amd.kabsch.shape_utils import Shape

    rs = np.random.default_rng(9)
    M = 5
    pcls = [np.concatenate([rs.uniform(-1, 1, (40, 3)), rs.uniform(0, 1, (40, 1))], -1).astype(np.float32) for _ in range(M)]
    boxes = Shape(pos=torch.zeros(M, 3), dims=torch.ones(M, 3) * 2, rot=torch.zeros(M, 1), probs=torch.ones(M, 1))
    sdb = BoxSnippetDb({"pcl_in_box_cosy": pcls, "boxes": boxes, "lidar_rows": [rs.integers(0, 64, 40).astype(np.uint8) for _ in range(M)]}, "cuda")
    box_cfg = {"max_num_objs": 4, "min_artificial_obj_velo": 1.0, "max_artificial_obj_velo": 3.0, "max_scale_delta": 0.2,
               "max_points_dropout": 0.25, "use_raydrop_augm": True}
    aug = BoxAugmenter(make_cfg(64, 40.0, box_cfg, network="pointpillars"), sdb, need_flow=False)
    pcl = torch.from_numpy(np.concatenate([rs.uniform(-5, 5, (200, 3)), rs.uniform(0, 1, (200, 1))], -1).astype(np.float32)).cuda()
    from liso_amd.datasets.torch_dataset_commons import voxelize_sample
    sample = {"pcl_ta": {"pcl": pcl, "pillar_coors": voxelize_sample(pcl, (40.0, 40.0), (64, 64))[0]}, "pcl_full_w_ground_ta": pcl,
              "pcl_full_no_ground_ta": pcl, "gt": {}}
    outs = []
    for _ in range(2):
        np.random.seed(3)
        torch.manual_seed(3)
        outs.append(aug.create_augmented_sample_from_box_snippet_db(0.1, sample))
    assert torch.equal(outs[0]["pcl_full_no_ground_ta"], outs[1]["pcl_full_no_ground_ta"])
    n_extra = outs[0]["pcl_full_no_ground_ta"].shape[0] - 200
    K = outs[0]["gt"]["boxes"].pos.shape[0]
    assert 1 <= K <= 4 and K <= n_extra <= 40 * K
    assert "centermaps_probs" not in outs[0]["mined"]


def test_fast_location_draws_place_objects_on_free_cells():
    """`reference_draws=False`: k distinct free cells, every centre at least the dilation radius away from occupied pillars"""
    from liso_amd.datasets.box_augmentation import BoxAugmenter, BoxSnippetDb
    from liso_amd.datasets.torch_dataset_commons import voxelize_sample
    from liso_amd.kabsch.shape_utils import Shape

    rs = np.random.default_rng(2)
    M, G, R = 20, 128, 60.0
    pcls = [np.concatenate([rs.uniform(-1, 1, (30, 3)), rs.uniform(0, 1, (30, 1))], -1).astype(np.float32) for _ in range(M)]
    boxes = Shape(pos=torch.zeros(M, 3), dims=torch.ones(M, 3) * 2, rot=torch.zeros(M, 1), probs=torch.ones(M, 1))
    box_cfg = {"max_num_objs": 15, "min_artificial_obj_velo": 1.0, "max_artificial_obj_velo": 3.0, "max_scale_delta": 0.2,
               "max_points_dropout": 0.25, "use_raydrop_augm": False}
    aug = BoxAugmenter(make_cfg(G, R, box_cfg, network="pointpillars"), BoxSnippetDb({"pcl_in_box_cosy": pcls, "boxes": boxes}, "cuda"),
                       need_flow=False, reference_draws=False)
    pcl = torch.from_numpy(np.concatenate([rs.uniform(-12, 12, (3000, 3)), rs.uniform(0, 1, (3000, 1))], -1).astype(np.float32)).cuda()
    coors = voxelize_sample(pcl, (R, R), (G, G))[0]
    sample = {"pcl_ta": {"pcl": pcl, "pillar_coors": coors}, "pcl_full_w_ground_ta": pcl, "pcl_full_no_ground_ta": pcl, "gt": {}}
    occupied = np.unique(coors.cpu().numpy(), axis=0)
    np.random.seed(1)
    torch.manual_seed(1)
    for _ in range(20):
        res = aug.create_augmented_sample_from_box_snippet_db(0.1, sample)
        pos = res["gt"]["boxes"].pos[:, :2]
        cells = voxelize_sample(torch.cat([pos, torch.zeros(pos.shape[0], 1)], -1), (R, R), (G, G))[0].numpy()
        assert len({tuple(c) for c in cells}) >= cells.shape[0] - 1  # distinct cells (the half-cell jitter may cross a border)
        d2 = ((cells[:, None, :] - occupied[None]) ** 2).sum(-1).min(-1)
        assert (d2 > (4 - 1.5) ** 2).all()  # outside the radius-4 disk of every occupied pillar, up to the half-cell jitter


def test_wrapper_carries_the_mined_boxes_along(golden_dir):
    """`create_augmented_sample_from_flow_cluster_detector_and_box_snippet_db` (reference :1805-1830): the sample's already-mined boxes are
    appended to the pasted ones; seeded like the direct call, it returns the same sample"""
    from liso_amd.datasets.box_augmentation import BoxAugmenter, BoxSnippetDb
    from liso_amd.kabsch.shape_utils import Shape

    g = np.load(f"{golden_dir}/box_augment_reference.npz")
    G, R, db, box_cfg, need_flow, seed, n_pre = load_case(g, "b")
    boxes = Shape(**{k: torch.from_numpy(g[f"b_db_box_{k}"]) for k in ("pos", "dims", "rot", "probs")})
    aug = BoxAugmenter(make_cfg(G, R, box_cfg, str(g["b_flow_source"])), BoxSnippetDb({"pcl_in_box_cosy": db["points"], "boxes": boxes}, "cuda"),
                       need_flow=need_flow)
    pcl = torch.from_numpy(g["b_in_pcl"]).cuda()
    pre = Shape(**{k: torch.from_numpy(g[f"b_pre_{k}"]) for k in ("pos", "dims", "rot", "probs", "velo")})
    sample = {"pcl_ta": {"pcl": pcl, "pillar_coors": torch.from_numpy(g["b_in_coors"]).cuda()}, "pcl_full_w_ground_ta": pcl,
              "pcl_full_no_ground_ta": pcl, "gt": {}, "mined": {"boxes": pre}}
    np.random.seed(seed)
    torch.manual_seed(seed)
    res = aug.create_augmented_sample_from_flow_cluster_detector_and_box_snippet_db(0.1, sample)
    assert np.array_equal(res["gt"]["boxes"].pos.numpy(), g["b_out_box_pos"])
    assert res["mined"]["prediscovered_boxes"].pos.shape[0] == n_pre == 3
    assert np.array_equal(res["pcl_ta"]["pillar_coors"].cpu().numpy(), g["b_out_coors"])
