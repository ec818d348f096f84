"""Generates tests/golden/tracker_reference.npz from the reference's own sequence trackers (liso/tracker/global_box_tracker.py:13-514
FlowBasedBoxTracker, liso/tracker/box_tracker.py:8-126 NotATracker) on synthetic sequences: objects on smooth paths seen from a
moving sensor, detections with noise, missed detections (holes of one frame that the tracker fills, longer gaps that end a track),
clutter, an empty frame.  Inputs and every result the mining loop reads (world / sensor boxes, track ids, attribute lists, longest
tracks, per-track box indices) are stored as flat arrays with per-frame offsets.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_tracker_golden.py
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_nms_iou_golden import install_native_stub  # noqa: E402
from make_targets_golden import _Anything, import_with_stubs  # noqa: E402

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")
ATTRS = ("pos", "dims", "rot", "probs", "velo", "class_id", "difficulty")


def pose(x, y, yaw, z=0.0):
    c, s = np.cos(yaw), np.sin(yaw)
    T = np.eye(4)
    T[:2, :2] = [[c, -s], [s, c]]
    T[:3, 3] = [x, y, z]
    return T


def sequence(seed, n_frames, n_objects, p_miss, n_clutter, empty_frame=None):
    """-> per frame: boxes in sensor coordinates, their poses propagated into the previous / next frame's sensor coordinates,
    odometry sensor(t) <- sensor(t+1), one integer attribute per box"""
    g = np.random.default_rng(seed)
    ego = [pose(0.8 * t, 0.05 * t * t, 0.01 * t) for t in range(n_frames + 1)]
    start = g.uniform(-30, 30, (n_objects, 2))
    vel = g.uniform(-1.5, 1.5, (n_objects, 2))
    dims = g.uniform([3.5, 1.6, 1.4], [5.0, 2.1, 1.9], (n_objects, 3))
    born, dies = g.integers(0, max(1, n_frames // 3), n_objects), g.integers(2 * n_frames // 3, n_frames + 1, n_objects)
    frames = []
    uid = 100
    for t in range(n_frames):
        rows = []
        for o in range(n_objects):
            if not (born[o] <= t < dies[o]) or g.uniform() < p_miss:
                continue
            here = [pose(*(start[o] + vel[o] * tt), np.arctan2(vel[o][1], vel[o][0])) for tt in (t - 1, t, t + 1)]
            noise = lambda: pose(*g.normal(0, 0.08, 2), g.normal(0, 0.01))  # noqa: E731
            s_T_w = np.linalg.inv(ego[t])
            det = s_T_w @ here[1] @ noise()
            into_prev = np.linalg.inv(ego[max(t - 1, 0)]) @ here[0] @ noise()
            into_next = np.linalg.inv(ego[min(t + 1, n_frames - 1)]) @ here[2] @ noise()
            rows.append((det, into_prev, into_next, dims[o], g.uniform(0.3, 1.0), o))
        for _ in range(g.integers(0, n_clutter + 1)):
            det = pose(*g.uniform(-35, 35, 2), g.uniform(-3, 3))
            rows.append((det, det @ pose(*g.normal(0, 0.3, 2), 0.0), det @ pose(*g.normal(0, 0.3, 2), 0.0),
                         g.uniform([3.5, 1.6, 1.4], [5.0, 2.1, 1.9]), g.uniform(0.1, 0.5), -1))
        if t == empty_frame:
            rows = []
        n = len(rows)
        det = np.stack([r[0] for r in rows]) if n else np.zeros((0, 4, 4))
        f = {"pos": det[:, :3, 3].astype(np.float32), "rot": np.arctan2(det[:, 1, 0], det[:, 0, 0])[:, None].astype(np.float32),
             "dims": np.stack([r[3] for r in rows]).astype(np.float32) if n else np.zeros((0, 3), np.float32),
             "probs": np.array([r[4] for r in rows], np.float32)[:, None],
             "into_prev": np.stack([r[1] for r in rows]) if n else np.zeros((0, 4, 4)),
             "into_next": np.stack([r[2] for r in rows]) if n else np.zeros((0, 4, 4)),
             "odom": np.linalg.inv(ego[t]) @ ego[t + 1], "attr": np.arange(uid, uid + n)}
        uid += n
        frames.append(f)
    return frames


def flat(out, key, arrays):
    out[key + "_offsets"] = np.cumsum([0] + [len(a) for a in arrays])
    out[key] = np.concatenate(arrays, axis=0) if len(arrays) else np.zeros((0,))


def shapes_to_flat(out, key, shapes):
    for a in ATTRS:
        vals = [getattr(s, a) for s in shapes]
        if vals[0] is None:
            continue
        flat(out, f"{key}_{a}", [v.detach().cpu().numpy() for v in vals])


def main():
    install_native_stub()

    def _imp():
        from liso.kabsch.shape_utils import Shape
        from liso.tracker.box_tracker import NotATracker
        from liso.tracker.global_box_tracker import FlowBasedBoxTracker
        return Shape, NotATracker, FlowBasedBoxTracker

    Shape, NotATracker, FlowBasedBoxTracker = import_with_stubs(_imp)
    cases = {"a": dict(seed=1, n_frames=14, n_objects=9, p_miss=0.18, n_clutter=2), "b": dict(seed=2, n_frames=9, n_objects=4, p_miss=0.3, n_clutter=1, empty_frame=4),
             "c": dict(seed=3, n_frames=25, n_objects=20, p_miss=0.12, n_clutter=3), "d": dict(seed=4, n_frames=6, n_objects=3, p_miss=0.0, n_clutter=0)}
    out = {"cases": np.array(sorted(cases))}
    for tag, kw in cases.items():
        frames = sequence(**kw)
        for k in ("pos", "rot", "dims", "probs", "into_prev", "into_next", "attr"):
            flat(out, f"{tag}_in_{k}", [f[k] for f in frames])
        out[f"{tag}_in_odom"] = np.stack([f["odom"] for f in frames])
        for name, make in (("flow", lambda: FlowBasedBoxTracker(use_propagated_boxes=True, box_matching_threshold_m=2.0)), ("none", NotATracker)):
            tr = make()
            for f in frames:
                n = len(f["pos"])
                boxes = Shape(pos=torch.from_numpy(f["pos"]), dims=torch.from_numpy(f["dims"]), rot=torch.from_numpy(f["rot"]),
                              probs=torch.from_numpy(f["probs"]), valid=torch.ones(n, dtype=torch.bool))
                tr.update(boxes, torch.from_numpy(f["into_next"]), torch.from_numpy(f["into_prev"]), torch.from_numpy(f["odom"]),
                          [{"uid": int(u)} for u in f["attr"]])
            tr.run_tracker()
            key = f"{tag}_{name}"
            shapes_to_flat(out, key + "_world", tr.get_boxes_in_world_coordinates())
            flat(out, key + "_ids", [t.numpy() for t in tr.track_ids])
            flat(out, key + "_attrs", [np.array([d["uid"] for d in fr], dtype=np.int64) for fr in tr.get_extra_attributes_at_each_timestamp()])
            shapes_to_flat(out, key + "_sensor", tr.get_boxes_in_sensor_coordinates_at_each_timestamp())
            ids, lens = tr.get_ids_lengths_of_longest_tracks()
            out[key + "_longest_lens"] = lens.numpy()
            out[key + "_id_set"] = np.sort(ids.numpy())
            lo, hi = tr.get_min_max_track_id()
            out[key + "_min_max"] = np.array([int(lo), int(hi)])
            out[key + "_counter"] = np.array(int(tr.max_track_id_counter))
            probe = ids[: min(5, len(ids))]
            out[key + "_probe_ids"] = probe.numpy()
            rows = [tr.get_box_indices_start_time_for_track_id(i) for i in probe]
            flat(out, key + "_probe_rows", [r[0].numpy() for r in rows])
            out[key + "_probe_start"] = np.array([int(r[1]) for r in rows])
            print(key, "frames", len(frames), "tracks", len(ids), "longest", lens[:5].tolist(),
                  "boxes", sum(len(f["pos"]) for f in frames), "->", sum(len(t) for t in tr.track_ids))
    np.savez_compressed(os.path.join(HERE, "tracker_reference.npz"), **out)


if __name__ == "__main__":
    sys.path.insert(0, "/root/reference")
    main()
