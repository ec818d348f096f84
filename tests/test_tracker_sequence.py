"""The sequence trackers of the mining loop (liso/tracker/global_box_tracker.py:13-514, box_tracker.py:8-126)
against tests/golden/tracker_reference.npz, written by the reference's own classes (tests/golden/make_tracker_golden.py): the
same detections, propagated poses and odometry must give the same tracks -- ids, hole-filling boxes, attribute lists, world / sensor
boxes, longest tracks, per-track box indices.  Host logic (a few dozen boxes per frame): runs without a GPU."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "tracker_reference.npz"))
ATTRS = ("pos", "dims", "rot", "probs")


def _frames(tag, key):
    off = G[f"{tag}_{key}_offsets"]
    return [G[f"{tag}_{key}"][off[i]:off[i + 1]] for i in range(len(off) - 1)]


def _run(tag, name):
    from liso_amd.kabsch.shape_utils import Shape
    from liso_amd.tracker.box_tracker import NotATracker
    from liso_amd.tracker.global_box_tracker import FlowBasedBoxTracker

    tr = FlowBasedBoxTracker(use_propagated_boxes=True, box_matching_threshold_m=2.0) if name == "flow" else NotATracker()
    ins = {k: _frames(tag, "in_" + k) for k in ("pos", "rot", "dims", "probs", "into_prev", "into_next", "attr")}
    for t in range(len(ins["pos"])):
        n = len(ins["pos"][t])
        boxes = Shape(pos=torch.from_numpy(ins["pos"][t]), dims=torch.from_numpy(ins["dims"][t]), rot=torch.from_numpy(ins["rot"][t]),
                      probs=torch.from_numpy(ins["probs"][t]), valid=torch.ones(n, dtype=torch.bool))
        tr.update(boxes, torch.from_numpy(ins["into_next"][t]), torch.from_numpy(ins["into_prev"][t]), torch.from_numpy(G[f"{tag}_in_odom"][t]),
                  [{"uid": int(u)} for u in ins["attr"][t]])
    tr.run_tracker()
    return tr


@pytest.mark.parametrize("name", ["flow", "none"])
@pytest.mark.parametrize("tag", [str(c) for c in G["cases"]])
def test_sequence_tracker_reproduces_the_reference(tag, name):
    tr = _run(tag, name)
    key = f"{tag}_{name}"
    ids = _frames(key, "ids")
    assert len(tr.track_ids) == len(ids)
    for t, want in enumerate(ids):
        assert np.array_equal(tr.track_ids[t].numpy(), want), (t, tr.track_ids[t], want)
    for view, boxes in (("world", tr.get_boxes_in_world_coordinates()), ("sensor", tr.get_boxes_in_sensor_coordinates_at_each_timestamp())):
        for a in ATTRS:
            for t, want in enumerate(_frames(key, f"{view}_{a}")):
                got = getattr(boxes[t], a).numpy()
                assert got.shape == want.shape and got.dtype == want.dtype, (view, a, t, got.shape, want.shape, got.dtype, want.dtype)
                assert np.allclose(got, want, rtol=0, atol=1e-9 if want.dtype == np.float64 else 1e-6), (view, a, t)
    for t, want in enumerate(_frames(key, "attrs")):
        assert [d["uid"] for d in tr.get_extra_attributes_at_each_timestamp()[t]] == want.tolist(), t
    tids, lens = tr.get_ids_lengths_of_longest_tracks()
    assert np.array_equal(lens.numpy(), G[key + "_longest_lens"]) and np.array_equal(np.sort(tids.numpy()), G[key + "_id_set"])
    lo, hi = tr.get_min_max_track_id()
    assert [int(lo), int(hi)] == G[key + "_min_max"].tolist() and int(tr.max_track_id_counter) == int(G[key + "_counter"])
    rows = _frames(key, "probe_rows")
    for i, tid in enumerate(G[key + "_probe_ids"]):
        box_idxs, start = tr.get_box_indices_start_time_for_track_id(int(tid))
        assert np.array_equal(box_idxs.numpy(), rows[i]) and int(start) == int(G[key + "_probe_start"][i])


def test_tracks_are_continuous_after_hole_filling():
    """property of the result, independent of the fixture: a track occupies consecutive frames, one box per frame"""
    tr = _run("c", "flow")
    seen = {}
    for t, ids in enumerate(tr.track_ids):
        assert len(set(ids.tolist())) == len(ids)
        for i in ids.tolist():
            seen.setdefault(i, []).append(t)
    assert all(ts == list(range(ts[0], ts[-1] + 1)) for ts in seen.values())
    assert max(len(ts) for ts in seen.values()) >= 20
