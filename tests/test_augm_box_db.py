"""The box-snippet database file utilities (liso_amd/tracker/augm_box_db_utils.py) against what the reference's own
save_augmentation_database / load_sanitize_box_augmentation_database write and return (tests/golden/make_augm_db_golden.py).
Host code only."""
import numpy as np
import torch

from liso_amd.kabsch.shape_utils import Shape
from liso_amd.tracker import augm_box_db_utils as u


def _build_db(g):
    db = u.get_empty_augm_box_db()
    for i in range(len(g["in_counts"])):
        db["pcl_in_box_cosy"].append(g[f"in_pcl_{i}"])
        db["lidar_rows"].append(g[f"in_rows_{i}"])
        db["boxes"].append(Shape(**{k: torch.from_numpy(g[f"in_box_{i}_{k}"]) for k in ("pos", "dims", "rot", "probs")}))
        db["box_T_sensor"].append(g[f"in_T_{i}"])
        db["unique_track_id"].append(i // 2)
    return db


def test_save_and_load_match_reference(golden_dir, tmp_path):
    g = np.load(f"{golden_dir}/augm_db_reference.npz")
    for tag, db in (("full", _build_db(g)), ("empty", u.get_empty_augm_box_db())):
        name, _ = u.save_augmentation_database(db, tmp_path / tag, 7)
        assert name.name == "boxes_db_global_step_7.npy"
        raw = np.load(name, allow_pickle=True).item()
        assert sorted(raw.keys()) == list(g[f"{tag}_saved_keys"])
        assert np.array_equal(raw["unique_track_id"], g[f"{tag}_saved_track_ids"]) and raw["unique_track_id"].dtype == np.uint32
        assert np.array_equal(raw["box_T_sensor"], g[f"{tag}_saved_T"])
        for k in ("pos", "dims", "rot", "probs", "valid"):
            assert np.array_equal(np.asarray(raw["boxes"][k]), g[f"{tag}_saved_box_{k}"]), (tag, k)
        loaded = u.load_sanitize_box_augmentation_database(name, 0.4)
        assert [p.shape[0] for p in loaded["pcl_in_box_cosy"]] == list(g[f"{tag}_loaded_counts"])
        assert np.array_equal(loaded["pcl_in_box_cosy"][0], g[f"{tag}_loaded_first_pcl"])
        for k in ("pos", "dims", "rot", "probs"):
            assert np.array_equal(getattr(loaded["boxes"], k).numpy(), g[f"{tag}_loaded_box_{k}"]), (tag, k)
        assert np.array_equal(loaded["box_T_sensor"].numpy(), g[f"{tag}_loaded_T"])
        assert len(loaded["lidar_rows"]) == len(loaded["pcl_in_box_cosy"])


def test_size_limit_drops_boxes(golden_dir):
    g = np.load(f"{golden_dir}/augm_db_reference.npz")
    db = _build_db(g)
    size = u.estimate_augm_db_size_mb(db)
    assert abs(size - sum(g["in_counts"]) * 16e-6) < 1e-12
    assert u.drop_boxes_from_augmentation_db(db, max_size_mb=1) is db  # below the limit: untouched
    np.random.seed(0)
    small = u.drop_boxes_from_augmentation_db(db, max_size_mb=size / 2)
    assert len(small["boxes"]) == int(len(db["boxes"]) / 2) == len(small["pcl_in_box_cosy"]) == len(small["unique_track_id"])
