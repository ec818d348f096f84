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

The bicycle-model variant (`smooth_track_bike_model`, :300-741; `track_smoothing_method: "bike_model"`) keeps the reference's optimiser
(torch.optim.LBFGS with strong-Wolfe line search, 20 inner iterations x `max_iters` steps) and replaces what every one of its loss
evaluations spends its time in: the scripted per-frame rollout (~30 launches per frame and direction) is one launch forward and one
backward (liso_bike_rollout_{fwd,bwd}_f32, one lane per track, hand-written adjoint)."""
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


def minimise_jerk_with_loss_history(observed_pos, valid_mask, max_iters=2000, learning_rate=0.1, pos_regul_loss_weight=3.0):
    """`return_losses=True` of the reference (:142-213): the same objective stepped by torch.optim.Adam one iteration at a time (the
    reference's formulation as tensor ops on the device), so that the three per-track loss terms of every iteration can be recorded --
    the one-launch kernel keeps its Adam moments in registers and records nothing.  A diagnostic path (the reference uses it for
    plots): ~25 launches per iteration.  -> (smoothed float32 [B,T,3], list of {per_batch_jerk_loss, per_batch_loss, pos_regul})"""
    obs = observed_pos.float()
    first = obs[:, :1].detach()
    rest = torch.nn.Parameter(obs[:, 1:, :].clone(), requires_grad=True)  # (the first position is fixed: BatchedSmoothTrack :36-64)
    opt = torch.optim.Adam([rest], lr=learning_rate)
    vm = valid_mask.float()
    n_valid = valid_mask.sum(dim=1)
    losses = []
    with torch.enable_grad():
        for _ in range(max_iters):
            opt.zero_grad()
            pos = torch.cat([first, rest], dim=1)
            jerk = torch.linalg.norm(torch.diff(pos, n=3, dim=1), dim=-1)
            jerk = torch.cat([jerk, jerk.new_zeros((jerk.shape[0], pos.shape[1] - jerk.shape[1]))], dim=1)
            jerk_loss = (jerk * vm).sum(dim=-1) / n_valid
            shift = ((pos - obs[:, :, :3]) ** 2).sum(dim=-1)
            regul = pos_regul_loss_weight * (shift * vm).sum(dim=-1) / n_valid
            per_track = jerk_loss + regul
            per_track.mean().backward()
            losses.append({"per_batch_jerk_loss": jerk_loss.detach().cpu().numpy(), "per_batch_loss": per_track.detach().cpu().numpy(),
                           "pos_regul": regul.detach().cpu().numpy()})
            opt.step()
    return torch.cat([first, rest.detach()], dim=1), losses


@torch.no_grad()
def smooth_track_jerk(batched_observed_pos_m, batched_valid_mask, batched_observed_yaw_angle_rad, time_between_frames_s: float,
                      pos_regul_loss_weight=3.0, max_iters=2000, learning_rate=0.1, verbose=False, return_losses=False):
    """same arguments and return tuple (smoothed positions, headings aligned with the direction of travel, per-frame displacement
    [, loss history with return_losses]) as the reference.  Like the reference, the headings are written INTO
    `batched_observed_yaw_angle_rad` and that tensor is returned."""
    if batched_observed_pos_m.shape[1] <= 4:  # min jerk needs more than 4 frames (the reference returns 3 values here either way)
        return batched_observed_pos_m, batched_observed_yaw_angle_rad, batched_displacement_from_pos(batched_observed_pos_m)
    losses = None
    if return_losses:
        track_positions_m, losses = minimise_jerk_with_loss_history(batched_observed_pos_m, batched_valid_mask, max_iters, learning_rate,
                                                                    pos_regul_loss_weight)
    else:
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
    if return_losses:
        return track_positions_m, rot_along_track, batched_displacement_from_pos(track_positions_m)[..., None], losses
    return track_positions_m, rot_along_track, batched_displacement_from_pos(track_positions_m)[..., None]


# ---- bicycle-model smoothing (reference :15-35, :300-741) ---------------------------------------------------------------------------
def torch_yaw_signed_diff(gt_yaw, pred_yaw, period: float = 2 * np.pi):
    diff = (gt_yaw - pred_yaw + period / 2) % period - period / 2
    return torch.where(diff > np.pi, diff - (2 * np.pi), diff)


def per_batch_mean_loss(per_element_loss, valid_mask, num_elements_in_batch):
    return torch.sum(per_element_loss * valid_mask.float(), dim=-1) / num_elements_in_batch


class _BikeRollout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, initial_state, accel, steering, vehicle_length, dt, max_yaw_rate, max_velocity):
        L.require_cuda(initial_state, accel, steering, vehicle_length)
        init, a, st = initial_state.float().contiguous(), accel.float().contiguous(), steering.float().contiguous()
        ln = vehicle_length.float().contiguous()
        B, T = a.shape
        states = torch.empty((B, T, 5), dtype=torch.float32, device=a.device)
        with torch.cuda.device(a.device):
            L.check(L.TIMER.launch("bike_rollout_fwd", lambda: L.lib().liso_bike_rollout_fwd_f32(
                B, T, L.ptr(init), L.ptr(a), L.ptr(st), L.ptr(ln), float(dt), float(max_yaw_rate), float(max_velocity), L.ptr(states),
                L.stream_ptr())), "bike_rollout_fwd")
        ctx.save_for_backward(a, st, ln, states)
        ctx.cfg = (float(dt), float(max_yaw_rate), float(max_velocity))
        return states

    @staticmethod
    def backward(ctx, grad_states):
        a, st, ln, states = ctx.saved_tensors
        B, T = a.shape
        g = grad_states.float().contiguous()
        gi = torch.empty((B, 5), dtype=torch.float32, device=a.device)
        ga, gs = torch.empty_like(a), torch.empty_like(st)
        dt, myr, mv = ctx.cfg
        with torch.cuda.device(a.device):
            L.check(L.TIMER.launch("bike_rollout_bwd", lambda: L.lib().liso_bike_rollout_bwd_f32(
                B, T, L.ptr(a), L.ptr(st), L.ptr(ln), dt, myr, mv, L.ptr(states), L.ptr(g), L.ptr(gi), L.ptr(ga), L.ptr(gs),
                L.stream_ptr())), "bike_rollout_bwd")
        return gi, ga, gs, None, None, None, None


class BatchedBikeModel(torch.nn.Module):
    """reference :340-457: a batch of tracks as kinematic bicycle models -- per-frame acceleration / steering inputs and the initial
    state are the parameters, the positions follow from the rollout.  state = [x, y, heading, velocity, heading rate]"""

    def __init__(self, *, batched_observed_track_pos, batched_vehicle_length, time_between_frames_s: float, max_yaw_rate: float,
                 max_velocity: float, optimize_initial_pos=True):
        super().__init__()
        assert len(batched_observed_track_pos.shape) == 3, batched_observed_track_pos.shape
        B, T, _ = batched_observed_track_pos.shape
        assert T >= MIN_TRACK_LEN_FOR_SMOOTHING, f"need at least {MIN_TRACK_LEN_FOR_SMOOTHING} positions for smoothing"
        self.batch_size, self.num_time_steps = int(B), int(T)
        obs = batched_observed_track_pos
        self.propagated_states = []
        velo0 = torch.linalg.norm(obs[:, 2:, :2] - obs[:, :-2, :2], dim=-1) / (2 * time_between_frames_s)
        yaw0 = get_orientations_along_track(pos=obs[:, :, :2])
        yaw_rate0 = torch_yaw_signed_diff(yaw0[:, 1:], yaw0[:, :1]) / time_between_frames_s
        P = torch.nn.Parameter
        self.accel_over_time = P(torch.zeros_like(obs[..., 0]), requires_grad=True)
        self.steering_input_over_time = P(torch.zeros_like(obs[..., 0]), requires_grad=True)
        self.initial_pos = P(obs[:, 0, 0:2], requires_grad=optimize_initial_pos)
        self.initial_yaw = P(yaw0[:, [0]], requires_grad=True)
        self.initial_velo_mps = P(velo0[:, [0]], requires_grad=True)
        self.initial_yaw_rate_radps = P(yaw_rate0[:, [0]], requires_grad=True)
        self.x_idx, self.y_idx, self.heading_idx, self.velo_idx, self.hdot_idx = 0, 1, 2, 3, 4
        self.time_between_frames_s = float(time_between_frames_s)
        assert batched_vehicle_length.shape == (B,)
        self.vehicle_length, self.max_yaw_rate, self.max_velocity = batched_vehicle_length, max_yaw_rate, max_velocity

    def forward(self):
        initial_state = torch.cat([self.initial_pos, self.initial_yaw, self.initial_velo_mps, self.initial_yaw_rate_radps], dim=-1)
        self.propagated_states = _BikeRollout.apply(initial_state, self.accel_over_time, self.steering_input_over_time, self.vehicle_length,
                                                    self.time_between_frames_s, self.max_yaw_rate, self.max_velocity)
        return self.propagated_states

    @property
    def pos(self):
        return self.propagated_states[..., [self.x_idx, self.y_idx]]

    @property
    def rot(self):
        return self.propagated_states[..., [self.heading_idx]]

    @property
    def yaw_rate(self):
        return self.propagated_states[..., [self.hdot_idx]]

    @property
    def velo(self):
        return self.propagated_states[..., [self.velo_idx]]

    def get_pos_jerk_magnitude(self):
        return torch.linalg.norm(torch.diff(self.pos, n=3, dim=0), dim=-1)


def smooth_track_bike_model(*, batched_observed_pos_m, batched_valid_mask, batched_observed_yaw_angle_rad, batched_vehicle_length_m,
                            time_between_frames_s: float, max_iters=30, learning_rate=0.1, accel_penalty_weight=0.1,
                            velo_penalty_weight=0.1, pos_regul_loss_weight=1.0, max_velocity_mps=50.0, max_yaw_rate_radps=np.pi / 2,
                            verbose=False, return_losses=False):
    """reference :577-741 -- same arguments and return tuple (positions [B,T,3], headings [B,T,1], per-frame displacement [B,T,1]
    (+ the loss terms of every evaluation with `return_losses`))"""
    batched_observed_pos_m = batched_observed_pos_m.clone()
    batched_observed_yaw_angle_rad = batched_observed_yaw_angle_rad.clone()
    if batched_observed_pos_m.shape[1] < MIN_TRACK_LEN_FOR_SMOOTHING:
        return batched_observed_pos_m, batched_observed_yaw_angle_rad, batched_displacement_from_pos(batched_observed_pos_m)
    track = BatchedBikeModel(batched_observed_track_pos=batched_observed_pos_m, time_between_frames_s=time_between_frames_s,
                             batched_vehicle_length=batched_vehicle_length_m, max_velocity=max_velocity_mps,
                             max_yaw_rate=max_yaw_rate_radps)
    optimizer = torch.optim.LBFGS(track.parameters(), lr=learning_rate, max_iter=20, line_search_fn="strong_wolfe")
    track.train()
    losses = []
    valid = batched_valid_mask
    n_valid = torch.sum(valid, dim=1)
    zeros = torch.zeros((valid.shape[0],), device=batched_observed_pos_m.device)

    def closure():
        optimizer.zero_grad()
        track.forward()
        lin = yaw = rate = zeros
        if accel_penalty_weight > 0.0:
            lin = accel_penalty_weight * per_batch_mean_loss(track.accel_over_time ** 2, valid, n_valid)
            yaw = accel_penalty_weight * per_batch_mean_loss(track.steering_input_over_time ** 2, valid, n_valid)
        if velo_penalty_weight > 0.0:
            r = torch.squeeze(track.yaw_rate, dim=-1)
            rate = per_batch_mean_loss(torch.where(torch.abs(r) > max_yaw_rate_radps, r ** 2, torch.zeros_like(r)), valid, n_valid)
        shift = ((track.pos - batched_observed_pos_m[:, :, :2]) ** 2).sum(dim=-1)
        pos = pos_regul_loss_weight * per_batch_mean_loss(shift, valid, n_valid)
        per_track = lin + yaw + rate + pos
        loss = per_track.mean()
        loss.backward()
        if return_losses:
            losses.append({"per_batch_loss": per_track.detach().cpu().numpy(), "linear_accel_penalty": lin.detach().cpu().numpy(),
                           "yaw_accel_penalty": yaw.detach().cpu().numpy(), "yaw_rate_penalty": rate.detach().cpu().numpy(),
                           "pos_regul": pos.detach().cpu().numpy()})
        return loss

    for _ in range(max_iters):
        optimizer.step(closure)
    optimized_pos = torch.cat([track.pos, batched_observed_pos_m[:, :, 2:]], dim=-1)
    optimized_velo = batched_displacement_from_pos(optimized_pos)[..., None]
    if return_losses:
        return optimized_pos, track.rot, optimized_velo, losses
    return optimized_pos, track.rot, optimized_velo
