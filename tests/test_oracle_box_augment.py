"""The CPU restatement of the box-snippet augmentation (oracle/box_augment.py) against the fixture generated from the reference's
own `create_augmented_sample_from_box_snippet_db` (tests/golden/make_box_augment_golden.py; the disk dilation in it was
evaluated by scikit-image): seeded like the fixture, every output must be identical."""
import numpy as np
import pytest
import torch

from oracle import box_augment as ob


def load_case(g, tag):
    G, R, max_objs, dropout, need_flow, seed, n_pre = g[f"{tag}_meta"]
    counts = g[f"{tag}_db_counts"]
    off = np.concatenate([[0], np.cumsum(counts)])
    pts = g[f"{tag}_db_points"]
    db = {"points": [pts[off[i]:off[i + 1]] for i in range(len(counts))], "dims": g[f"{tag}_db_box_dims"], "pos_z": g[f"{tag}_db_box_pos"][:, 2]}
    box_cfg = {"max_num_objs": int(max_objs), "min_artificial_obj_velo": 1.0, "max_artificial_obj_velo": 3.0, "max_scale_delta": 0.2,
               "max_points_dropout": float(dropout), "use_raydrop_augm": False}
    return int(G), float(R), db, box_cfg, bool(need_flow), int(seed), int(n_pre)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_reproduces_the_reference_sample(golden_dir, tag):
    g = np.load(f"{golden_dir}/box_augment_reference.npz")
    G, R, db, box_cfg, need_flow, seed, n_pre = load_case(g, tag)
    np.random.seed(seed)
    torch.manual_seed(seed)
    o = ob.augment(g[f"{tag}_in_pcl"], g[f"{tag}_in_coors"], g[f"{tag}_in_flow"], db, [R, R], [G, G], box_cfg, need_flow=need_flow)
    assert np.array_equal(o["free_mask"], g[f"{tag}_free_mask"])
    assert ob.dilation_radius([R, R], [G, G]) == int(g[f"{tag}_radius"])
    K = o["box_pos"].shape[0]
    assert K + n_pre == g[f"{tag}_out_box_pos"].shape[0]
    for k in ("pos", "dims", "rot", "velo"):
        assert np.array_equal(o[f"box_{k}"], g[f"{tag}_out_box_{k}"][:K]), k
    assert np.array_equal(o["pcl"], g[f"{tag}_out_pcl"])
    assert np.array_equal(o["pillar_coors"], g[f"{tag}_out_coors"])
    n_in = g[f"{tag}_in_pcl"].shape[0]
    assert np.array_equal(o["extra_pcl"], g[f"{tag}_out_pcl_full_no_ground"][n_in:])
    if need_flow:
        assert np.array_equal(o["flow"], g[f"{tag}_out_flow"])
    if tag == "b":
        assert not o["in_range"].all()  # this case has pasted points outside the BEV range: the drop is exercised


def test_disk_and_mask_edge_cases():
    assert ob.disk(3).sum() == 29 and ob.disk(1).sum() == 5  # skimage.morphology.disk
    # an occupied corner cell blocks exactly the quarter disk; an empty sweep leaves everything free
    m = ob.free_location_mask(np.array([[0, 0]]), (16, 16), 3)
    assert int((~m).sum()) == 11 and not m[0, 3] and m[0, 4] and not m[2, 2] and m[3, 1]
    assert ob.free_location_mask(np.zeros((0, 2), np.int64), (8, 8), 3).all()
