"""Minimum-jerk track smoothing.  Mirror of liso/tracker/track_smoothing.py: `smooth_track_jerk` (:104-290, the
tracker's default `track_smoothing_method`), `batched_displacement_from_pos` (:87-101), `get_orientations_along_track` (:460-487),
`is_far_enough_for_rot_alignment` (:293-297), `MIN_TRACK_LEN_FOR_SMOOTHING`.

The reference optimises a [B, T, 3] tensor with 2000 autograd + Adam steps (≈25 small launches each; 1.8 s per batch on the host).  Here
the whole optimisation is ONE kernel launch (include/liso_tracking.h: liso_smooth_tracks_jerk_f32, one block per track, positions in
LDS, analytic gradient, torch.optim.Adam's update order in fp32); the heading alignment afterwards is a handful of tensor ops on the
device without a host read (the reference's `while not all aligned` loop runs its full, bounded number of rounds: rounds after
everything is aligned change nothing).

Parity: the reference's own result moves by 4e-2 m when one input coordinate changes by 1e-6 (Adam at lr 0.1 without decay keeps
jittering around the optimum; measured, stored in tests/golden/track_smoothing_reference.npz).  Short runs (1 / 3 / 20 steps) match the
reference to float32 rounding; the 2000-step result matches within that sensitivity and in the final loss.

The bicycle-model variant (`smooth_track_bike_model`) is not built."""
import numpy as np
import torch

from liso_amd import _lib as L

MIN_TRACK_LEN_FOR_SMOOTHING = 4


def batched_displacement_from_pos(pos, num_skip=1):
    assert num_skip >= 1, num_skip
    disp = torch.linalg.norm(pos[:, num_skip:, :] - pos[:, :-num_skip, :], dim=-1)
    if num_skip == 1:
        disp = torch.cat([disp, disp[:, [-num_skip]]], dim=1)
    else:
        disp = torch.cat([disp[:, :(num_skip // 2)], disp, disp[:, (-num_skip // 2):]], dim=1)
    assert pos.shape[:-1] == disp.shape, (pos.shape, disp.shape)
    return disp


def get_orientations_along_track(pos, pad_borders=True, num_skip=2):
    dir_vecs = (pos[:, num_skip:, :2] - pos[:, :-num_skip, :2]).detach()
    dir_vecs = dir_vecs / torch.linalg.norm(dir_vecs, dim=-1, keepdim=True).clamp_min(0.00001)
    track_angle = torch.atan2(dir_vecs[:, :, 1], dir_vecs[:, :, 0])
    if pad_borders:
        if num_skip == 1:
            track_angle = torch.cat([track_angle, track_angle[:, [-num_skip]]], dim=1)
        else:
            track_angle = torch.cat([track_angle[:, :(num_skip // 2)], track_angle, track_angle[:, (-num_skip // 2):]], dim=1)
        assert pos.shape[:-1] == track_angle.shape, (pos.shape, track_angle.shape)
    return track_angle


def is_far_enough_for_rot_alignment(track_displacement_m, min_disp_for_rot_alignment_m: float):
    return track_displacement_m > min_disp_for_rot_alignment_m


@torch.no_grad()
def minimise_jerk(observed_pos, valid_mask, max_iters=2000, learning_rate=0.1, pos_regul_loss_weight=3.0):
    """the Adam loop of the reference (:126-213) as one launch: float32 [B,T,3] -> smoothed float32 [B,T,3]"""
    L.require_cuda(observed_pos)
    obs = observed_pos.float().contiguous()
    B, T, C = obs.shape
    assert C == 3 and valid_mask.shape == (B, T), (obs.shape, valid_mask.shape)
    out = torch.empty_like(obs)
    val = valid_mask.to(torch.uint8).contiguous()
    with torch.cuda.device(obs.device):
        L.check(L.TIMER.launch("smooth_tracks_jerk", lambda: L.lib().liso_smooth_tracks_jerk_f32(
            L.ptr(obs), L.ptr(val), B, T, int(max_iters), float(learning_rate), float(pos_regul_loss_weight), L.ptr(out),
            L.stream_ptr())), "smooth_tracks_jerk")
    return out


@torch.no_grad()
def smooth_track_jerk(batched_observed_pos_m, batched_valid_mask, batched_observed_yaw_angle_rad, time_between_frames_s: float,
                      pos_regul_loss_weight=3.0, max_iters=2000, learning_rate=0.1, verbose=False, return_losses=False):
    """same arguments and return tuple (smoothed positions, headings aligned with the direction of travel, per-frame displacement) as
    the reference.  Like the reference, the headings are written INTO `batched_observed_yaw_angle_rad` and that tensor is returned."""
    if return_losses:
        raise NotImplementedError("the per-iteration loss history is not recorded by the one-launch optimisation")
    if batched_observed_pos_m.shape[1] <= 4:  # min jerk needs more than 4 frames
        return batched_observed_pos_m, batched_observed_yaw_angle_rad, batched_displacement_from_pos(batched_observed_pos_m)
    track_positions_m = minimise_jerk(batched_observed_pos_m, batched_valid_mask, max_iters, learning_rate, pos_regul_loss_weight)
    rot_along_track = batched_observed_yaw_angle_rad.detach()
    aligned = ~batched_valid_mask
    for num_skip in range(1, min(10, track_positions_m.shape[1] // 2) + 1):
        far = is_far_enough_for_rot_alignment(batched_displacement_from_pos(track_positions_m, num_skip=num_skip), 1.0)
        can = ~aligned & far
        angles = get_orientations_along_track(track_positions_m, pad_borders=True, num_skip=num_skip)[..., None]
        rot_along_track.copy_(torch.where(can[..., None], angles.to(rot_along_track.dtype), rot_along_track))
        aligned = aligned | can
    rot_along_track[:, 0, :] = rot_along_track[:, 1, :]  # constant heading at both ends of the track
    batch_idx = torch.arange(batched_valid_mask.shape[0], device=rot_along_track.device)
    last = batched_valid_mask.sum(dim=1) - 1
    rot_along_track[batch_idx, last, 0] = rot_along_track[batch_idx, last - 1, 0]
    return track_positions_m, rot_along_track, batched_displacement_from_pos(track_positions_m)[..., None]
