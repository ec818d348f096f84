"""The box-snippet database on disk: mirror of liso/tracker/augm_box_db_utils.py (same function names, same `.npy` layout -- a pickled
dict with `pcl_in_box_cosy` (list of [n_i, 4] float32), `lidar_rows` (list), `boxes` (the attribute dict of a numpy Shape),
`box_T_sensor` [M, 4, 4] and `unique_track_id` [M] uint32 -- so databases written by either implementation load in the other).
`liso_amd.datasets.box_augmentation.BoxSnippetDb` takes the loaded dictionary to the device."""
import os
from copy import deepcopy
from datetime import datetime
from pathlib import Path
from typing import Dict, List, Union

import numpy as np
import torch

from liso_amd.kabsch.shape_utils import Shape

MIN_POINTS_PER_BOX = 10  # reference :19: snippets with at most this many points are dropped at load time


def load_sanitize_box_augmentation_database(path_to_augmentation_db: Union[str, Path], confidence_threshold_mined_boxes: float):
    """reference :13-59: keep the snippets with more than 10 points whose box confidence reaches the threshold"""
    print(f"Loading augmentation boxes from db at {path_to_augmentation_db}")
    db = np.load(path_to_augmentation_db, allow_pickle=True).item()
    num_pts = np.array([p.shape[0] for p in db["pcl_in_box_cosy"]])
    confident = np.squeeze(db["boxes"]["probs"], axis=-1) >= confidence_threshold_mined_boxes
    keep = (num_pts > MIN_POINTS_PER_BOX) & confident
    print(f"Dropping {np.count_nonzero(~confident)}/{confident.shape[0]} boxes from augmentation db - they are not more confident "
          f"than {confidence_threshold_mined_boxes}!")
    keep_t = torch.from_numpy(keep)
    db["pcl_in_box_cosy"] = [p for p, k in zip(db["pcl_in_box_cosy"], keep) if k]
    db["lidar_rows"] = [r for r, k in zip(db["lidar_rows"], keep) if k]
    db["boxes"] = Shape(**db["boxes"]).to_tensor()[keep_t]
    db["box_T_sensor"] = torch.from_numpy(db["box_T_sensor"])[keep_t]
    assert db["box_T_sensor"].shape[0] == db["boxes"].shape[0] == len(db["pcl_in_box_cosy"]), (
        db["box_T_sensor"].shape, db["boxes"].shape, len(db["pcl_in_box_cosy"]))
    print(f"Loaded {sum(db['boxes'].shape)} augmentation boxes from db at {path_to_augmentation_db}")
    return db


def get_empty_augm_box_db():
    return {"pcl_in_box_cosy": [], "lidar_rows": [], "boxes": [], "box_T_sensor": [], "unique_track_id": []}


def estimate_augm_db_size_mb(db):
    return sum(v.nbytes for v in db["pcl_in_box_cosy"]) * 1e-6


def drop_boxes_from_augmentation_db(db: Dict[str, List], max_size_mb: int):
    """reference :78-110: shrink the database to `max_size_mb` of point data -- randomly when all confidences are equal, else by
    raising a confidence floor in steps of 0.001"""
    before = estimate_augm_db_size_mb(db)
    if before <= max_size_mb:
        return db
    conf = np.squeeze(np.stack([box.probs for box in db["boxes"]]), axis=-1)
    num_keep = int(len(db["boxes"]) / (before / max_size_mb))
    # (the reference tests `len(np.unique(conf) == 1)`, the length of a boolean array, which is true for any non-empty database: the
    # random branch is the one that runs; kept)
    if len(np.unique(conf) == 1):
        keep_idxs = np.random.choice(np.arange(0, len(db["boxes"])), num_keep, replace=False)
    else:
        floor, keep = conf.min(), np.ones_like(conf, dtype=bool)
        while keep.sum() > num_keep:
            floor = floor + 0.001
            keep[conf < floor] = False
        keep_idxs = np.arange(0, len(db["boxes"]))[keep]
    small = {k: [v[i] for i in keep_idxs] for k, v in db.items()}
    print(f"{datetime.now().strftime('%Y%m%d_%H%M%S')}: Dropped from {before}Mb to {estimate_augm_db_size_mb(small)}Mb from db!")
    return small


def save_augmentation_database(db, export_raw_tracked_detections_to: Path, global_step: int):
    """reference :113-188: stack the per-snippet entries and write `boxes_db_global_step_<step>.npy`; an empty database is written as
    the reference's one-box placeholder (11 points, so that it survives the loader's point filter)"""
    out = {k: deepcopy(v) for k, v in db.items()}
    target = Path(export_raw_tracked_detections_to)
    target.mkdir(exist_ok=True, parents=True)
    if len(out["box_T_sensor"]) == 0:
        print("WARNING: Not a single object was mined! Writing the placeholder database.")
        out["unique_track_id"] = np.array([0], dtype=np.uint32)
        out["box_T_sensor"] = np.eye(4, dtype=np.float64)[None]
        out["boxes"] = Shape(pos=np.array([10.0, 0.0, 0.0]), dims=np.array([10.0, 5.0, 1.0]), rot=np.array([0.0]), probs=np.array([1.0]),
                             velo=np.array([1.0]))[None].__dict__
        pts = np.array([[2.0, 3.0, 1.0, 1.0], [3.0, 3.0, 1.0, 1.0], [4.0, -3.0, 1.0, 1.0], [5.0, 1.0, 1.0, 1.0]] + [[2.0, 3.0, 1.0, 1.0]] * 7,
                       dtype=np.float32)
        out["lidar_rows"] = [np.arange(pts.shape[0], dtype=np.uint8)]
        out["pcl_in_box_cosy"] = [pts]
    else:
        out["unique_track_id"] = np.stack(out["unique_track_id"], axis=0).astype(np.uint32)
        out["box_T_sensor"] = np.stack(out["box_T_sensor"], axis=0)
        out["boxes"] = Shape.from_list_of_shapes(out["boxes"]).cpu().numpy().__dict__
    save_name = target / f"boxes_db_global_step_{global_step}.npy"
    np.save(save_name, out)
    size_in_mb = os.path.getsize(save_name) >> 20
    print(f"Saving {len(out['pcl_in_box_cosy'])} boxes ({size_in_mb} Mb) with point clouds to {save_name}")
    return save_name, size_in_mb
