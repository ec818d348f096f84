"""The CPU restatement of the tracker's local box refinement (oracle/box_refinement.py) against the fixture written by the reference's
own `fit_2d_box_modest` / `perform_local_box_refinement` (tests/golden/make_box_refinement_golden.py)."""
import numpy as np
import pytest

from oracle import box_refinement as ob


def load_track(g, tag):
    age, start, fit_rot, fit_pos, flow_cluster = g[f"{tag}_meta"]
    sizes = g[f"{tag}_cloud_sizes"]
    off = np.concatenate([[0], np.cumsum(sizes)])
    clouds = [g[f"{tag}_clouds"][off[i]:off[i + 1]] for i in range(len(sizes))]
    return int(age), int(start), bool(fit_rot), bool(fit_pos), (0.95 if flow_cluster else 0.6), clouds


def test_rectangle_fit_matches_reference(golden_dir):
    g = np.load(f"{golden_dir}/box_refinement_reference.npz")
    for i in range(5):
        center, length, width, yaw = ob.closeness_fit(g[f"fit{i}_points"][:, :2])
        want = g[f"fit{i}_result"]
        assert np.allclose([center[0], center[1], length, width, yaw], want, rtol=0, atol=1e-12), (i, want)


@pytest.mark.parametrize("tag", ["t0", "t1", "t2"])
def test_track_refinement_matches_reference(golden_dir, tag):
    g = np.load(f"{golden_dir}/box_refinement_reference.npz")
    age, start, fit_rot, fit_pos, q, clouds = load_track(g, tag)
    pos, dims, rot = ob.perform_local_box_refinement(clouds, g[f"{tag}_in_pos"], g[f"{tag}_in_dims"], g[f"{tag}_in_rot"], age, start,
                                                     fit_rot, fit_pos, 1.2, q)
    assert np.allclose(rot, g[f"{tag}_out_rot"], rtol=0, atol=1e-6)
    assert np.allclose(dims, g[f"{tag}_out_dims"], rtol=0, atol=1e-6)
    assert np.allclose(pos, g[f"{tag}_out_pos"], rtol=0, atol=2e-6)
