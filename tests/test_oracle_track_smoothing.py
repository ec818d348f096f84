"""The CPU restatement of the minimum-jerk track smoothing (oracle/track_smoothing.py) against the fixture written by the reference's
`smooth_track_jerk`: tight for short runs; for the full 2000 steps within the reference's own sensitivity to a 1e-6 input change
(4e-2 m, stored in the fixture) and on the final loss."""
import numpy as np
import pytest

from oracle import track_smoothing as ot


@pytest.mark.parametrize("tag", ["a", "b"])
def test_short_runs_match_reference_tightly(golden_dir, tag):
    g = np.load(f"{golden_dir}/track_smoothing_reference.npz")
    obs, valid, yaw = g[f"{tag}_pos"], g[f"{tag}_valid"], g[f"{tag}_yaw"]
    for iters, tol in ((1, 1e-5), (3, 2e-5), (20, 1e-3)):  # (positions up to 75 m: one float32 ulp is 8e-6)
        p, rot, velo = ot.smooth_track_jerk(obs, valid, yaw, iters)
        assert np.abs(p - g[f"{tag}_{iters}_pos"]).max() <= tol, (iters, np.abs(p - g[f"{tag}_{iters}_pos"]).max())
        assert np.abs(velo - g[f"{tag}_{iters}_velo"])[valid].max() <= 10 * tol
        assert np.abs(rot - g[f"{tag}_{iters}_rot"])[valid].max() <= 1e-3


@pytest.mark.parametrize("tag", ["a", "b"])
def test_full_run_matches_within_the_reference_sensitivity(golden_dir, tag):
    g = np.load(f"{golden_dir}/track_smoothing_reference.npz")
    obs, valid, yaw = g[f"{tag}_pos"], g[f"{tag}_valid"], g[f"{tag}_yaw"]
    assert 0.01 < float(g[f"{tag}_sensitivity"]) < 0.2
    p, rot, velo = ot.smooth_track_jerk(obs, valid, yaw, 2000)
    assert np.abs(p - g[f"{tag}_2000_pos"])[valid].max() <= 3 * float(g[f"{tag}_sensitivity"])
    total, jerk = ot.losses(p, obs, valid)
    assert np.allclose(total, g[f"{tag}_2000_last_loss"], rtol=0.05), (total, g[f"{tag}_2000_last_loss"])
    # the objective really went down: far below the loss after one step
    assert (total < 0.7 * g[f"{tag}_1_last_loss"]).all()


def test_short_tracks_are_returned_unchanged(golden_dir):
    g = np.load(f"{golden_dir}/track_smoothing_reference.npz")
    p, rot, velo = ot.smooth_track_jerk(g["c_pos"], g["c_valid"], g["c_yaw"], 20)
    assert np.array_equal(p, g["c_20_pos"]) and np.array_equal(rot, g["c_20_rot"]) and np.allclose(velo, g["c_20_velo"], atol=1e-6)
