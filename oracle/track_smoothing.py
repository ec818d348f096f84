"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product may not).

CPU restatement (numpy float32, analytic gradient) of liso/tracker/track_smoothing.py:104-290 smooth_track_jerk: the Adam loop on the
minimum-jerk objective, then the alignment of the headings with the direction of travel.  Pinned by
tests/golden/track_smoothing_reference.npz (written by the reference's function).  The reference's 2000-step result moves by 4e-2 m
under a 1e-6 input change (stored in the fixture), so only short runs can be compared tightly."""
import numpy as np

F = np.float32


def adam_jerk(obs, valid, iters, lr=0.1, w_reg=3.0):
    obs = np.asarray(obs, F)
    B, T, _ = obs.shape
    p = obs.copy()
    mask = valid.astype(F)
    n = mask.sum(1, keepdims=True)
    m1, m2 = np.zeros_like(p), np.zeros_like(p)
    for it in range(1, iters + 1):
        d = np.zeros_like(p)
        d[:, :T - 3] = ((p[:, 3:] - F(3) * p[:, 2:-1]) + F(3) * p[:, 1:-2]) - p[:, :-3]
        nrm = np.sqrt((d * d).sum(-1, keepdims=True, dtype=F))
        u = np.where(nrm > 0, d * (mask[..., None] / np.where(nrm > 0, nrm, F(1))), F(0)).astype(F)
        gj = -u
        gj[:, 1:] += F(3) * u[:, :-1]
        gj[:, 2:] -= F(3) * u[:, :-2]
        gj[:, 3:] += u[:, :-3]
        g = ((gj + F(w_reg) * F(2) * mask[..., None] * (p - obs)) / (F(B) * n[..., None])).astype(F)
        g[:, 0] = 0
        m1 = (F(0.9) * m1 + F(0.1) * g).astype(F)
        m2 = (F(0.999) * m2 + F(0.001) * g * g).astype(F)
        bc1, bc2 = F(1 - 0.9 ** it), F(np.sqrt(1 - 0.999 ** it))
        step = (F(lr) / bc1) * (m1 / (np.sqrt(m2) / bc2 + F(1e-8)))
        step[:, 0] = 0
        p = (p - step).astype(F)
    return p


def losses(p, obs, valid, w_reg=3.0):
    """per-track total and jerk loss of positions p (the quantities the reference reports with return_losses)"""
    mask = valid.astype(np.float64)
    n = mask.sum(1)
    d = np.zeros(p.shape[:2])
    d[:, :p.shape[1] - 3] = np.linalg.norm(np.diff(p.astype(np.float64), n=3, axis=1), axis=-1)
    jerk = (d * mask).sum(1) / n
    reg = w_reg * ((((p - obs).astype(np.float64)) ** 2).sum(-1) * mask).sum(1) / n
    return jerk + reg, jerk


def displacement(pos, num_skip=1):
    """batched_displacement_from_pos (:87-101)"""
    disp = np.linalg.norm(pos[:, num_skip:] - pos[:, :-num_skip], axis=-1)
    if num_skip == 1:
        return np.concatenate([disp, disp[:, [-1]]], 1)
    return np.concatenate([disp[:, :num_skip // 2], disp, disp[:, (-num_skip // 2):]], 1)


def orientations(pos, num_skip):
    """get_orientations_along_track with pad_borders (:460-487)"""
    dv = pos[:, num_skip:, :2] - pos[:, :-num_skip, :2]
    dv = dv / np.maximum(np.linalg.norm(dv, axis=-1, keepdims=True), F(0.00001))
    ang = np.arctan2(dv[..., 1], dv[..., 0])
    if num_skip == 1:
        return np.concatenate([ang, ang[:, [-1]]], 1)
    return np.concatenate([ang[:, :num_skip // 2], ang, ang[:, (-num_skip // 2):]], 1)


def align_rotations(pos, yaw, valid, min_disp=1.0):
    """:232-279: headings follow the direction of travel wherever the track moved far enough over a growing frame gap; the first
    frame copies the second, the last valid frame the one before it"""
    rot = yaw.copy()
    aligned = ~valid
    k = 0
    while not aligned.all() and k < min(10, pos.shape[1] // 2):
        k += 1
        can = ~aligned & (displacement(pos, k) > min_disp)
        rot[can] = orientations(pos, k)[..., None][can]
        aligned |= can
    rot[:, 0] = rot[:, 1]
    last = valid.sum(1) - 1
    idx = np.arange(pos.shape[0])
    rot[idx, last, 0] = rot[idx, last - 1, 0]
    return rot


def smooth_track_jerk(obs, valid, yaw, iters=2000, lr=0.1, w_reg=3.0):
    if obs.shape[1] <= 4:
        return obs, yaw, displacement(obs)
    p = adam_jerk(obs, valid, iters, lr, w_reg)
    return p, align_rotations(p, yaw, valid), displacement(p)[..., None]
