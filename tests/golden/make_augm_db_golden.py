"""Generates tests/golden/augm_db_reference.npz with the reference's own liso/tracker/augm_box_db_utils.py: a database of 7 snippets
written by `save_augmentation_database` and read back by `load_sanitize_box_augmentation_database` (threshold 0.4), plus the
placeholder database the reference writes when nothing was mined.  Stored: the inputs, the raw bytes-independent contents of the
saved dictionaries, and what the loader returns.  Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_augm_db_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import import_with_stubs  # noqa: E402


def main():
    def _imp():
        import liso.tracker.augm_box_db_utils as u
        from liso.kabsch.shape_utils import Shape
        return u, Shape

    u, Shape = import_with_stubs(_imp)
    g = np.random.default_rng(5)
    out = {}
    db = u.get_empty_augm_box_db()
    counts = [30, 8, 55, 11, 10, 200, 12]
    probs = [0.9, 0.9, 0.3, 0.5, 0.8, 0.45, 0.39]
    for i, (n, p) in enumerate(zip(counts, probs)):
        db["pcl_in_box_cosy"].append(g.normal(size=(n, 4)).astype(np.float32))
        db["lidar_rows"].append(g.integers(0, 64, n).astype(np.uint8))
        db["boxes"].append(Shape(pos=torch.tensor(g.normal(size=3), dtype=torch.float32), dims=torch.tensor(g.uniform(1, 5, 3), dtype=torch.float32),
                                 rot=torch.tensor(g.normal(size=1), dtype=torch.float32), probs=torch.tensor([p], dtype=torch.float32)))
        db["box_T_sensor"].append(np.linalg.inv(np.eye(4) + 0.1 * g.normal(size=(4, 4))))
        db["unique_track_id"].append(i // 2)
        out[f"in_pcl_{i}"], out[f"in_rows_{i}"] = db["pcl_in_box_cosy"][-1], db["lidar_rows"][-1]
        for k in ("pos", "dims", "rot", "probs"):
            out[f"in_box_{i}_{k}"] = getattr(db["boxes"][-1], k).numpy()
        out[f"in_T_{i}"] = db["box_T_sensor"][-1]
    out["in_counts"], out["in_probs"] = np.array(counts), np.array(probs)
    with tempfile.TemporaryDirectory() as tmp:
        for tag, d in (("full", db), ("empty", u.get_empty_augm_box_db())):
            name, _ = u.save_augmentation_database(d, tmp, 7)
            raw = np.load(name, allow_pickle=True).item()
            out[f"{tag}_saved_keys"] = np.array(sorted(raw.keys()))
            out[f"{tag}_saved_track_ids"] = raw["unique_track_id"]
            out[f"{tag}_saved_T"] = raw["box_T_sensor"]
            for k, v in raw["boxes"].items():
                if v is not None:
                    out[f"{tag}_saved_box_{k}"] = np.asarray(v)
            loaded = u.load_sanitize_box_augmentation_database(name, 0.4)
            out[f"{tag}_loaded_counts"] = np.array([p.shape[0] for p in loaded["pcl_in_box_cosy"]])
            out[f"{tag}_loaded_first_pcl"] = loaded["pcl_in_box_cosy"][0]
            for k in ("pos", "dims", "rot", "probs"):
                out[f"{tag}_loaded_box_{k}"] = getattr(loaded["boxes"], k).numpy()
            out[f"{tag}_loaded_T"] = loaded["box_T_sensor"].numpy()
    np.savez_compressed(os.path.join(HERE, "augm_db_reference.npz"), **out)
    print({k: v.shape for k, v in out.items() if "loaded" in k})


if __name__ == "__main__":
    main()
