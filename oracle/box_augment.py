"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product may not).

CPU restatement (numpy + scipy.ndimage) of the reference's box-snippet augmentation,
    liso/datasets/torch_dataset_commons.py:1531-1776  create_augmented_sample_from_box_snippet_db
in the configuration the reference ships (liso_config.yml:56-66: use_raydrop_augm False, max_points_dropout 0.25), up to and
including the re-pillarisation (`pillarize_bev` :1147-1163 with `voxelize_sample` :975-987) -- the target maps of the result are
covered by their own fixture (targets_reference.npz).  It consumes numpy's and torch's global generators in the reference's order,
so seeded runs reproduce the reference's outputs: pinned by tests/golden/box_augment_reference.npz (generated from the reference
method itself, tests/golden/make_box_augment_golden.py).

Reference quirks that are kept on purpose:
  * the pasted points are NOT shifted in z: `torch_compose_matrix(..., t_z=None)` (:1582-1587) builds the box pose with zero z
    translation, while the box that is reported gets `box_z_pos_new` (:1573-1579);
  * the xy jitter is added in float64 and rounded back into the float32 location tensor (:1563-1566);
  * sin / cos of the heading are evaluated in float32 and widened (torch_transformation.py:37-53).
"""
import numpy as np
import torch
from scipy import ndimage as ndi


def disk(radius):
    """skimage.morphology.disk: (x^2 + y^2) <= r^2 on the (2r+1)^2 grid"""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2) <= radius ** 2


def dilation_radius(bev_range_m, grid, min_dist_m=2.0):
    """:1544-1553"""
    ppm = (np.array(grid).astype(np.int32) / np.array(bev_range_m, np.float32)).astype(np.float32)
    return max(3, int(min_dist_m / (1 / ppm).mean()))


def free_location_mask(pillar_coors, grid, radius):
    """True where an object centre may be placed: no occupied pillar within the disk (:1538-1557; skimage's binary_dilation is
    scipy.ndimage.binary_dilation with the footprint as structure, border value 0)"""
    occ = np.zeros(tuple(grid), bool)
    occ[pillar_coors[:, 0], pillar_coors[:, 1]] = True
    return ~ndi.binary_dilation(occ, structure=disk(radius))


def bev_center_coords(bev_range_m, grid):
    """liso/utils/bev_utils.py:5-38 + :45-49: metric centre of BEV cell (i, j), float64 arithmetic, stored float32 [H, W, 2]"""
    r = np.array(bev_range_m, np.float32)
    gs = np.array(grid).astype(np.int32)
    ext = 0.5 * np.array([-r[0], -r[1], r[0], r[1]])
    c = np.stack(np.meshgrid(np.arange(gs[0]), np.arange(gs[1]), indexing="ij"), axis=-1) + 0.5
    c /= gs
    c *= ext[2:] - ext[:2]
    c += ext[:2]
    return c.astype(np.float32)


def voxelize_sample(pcl, bev_range_m, grid):
    """:975-987 with liso/datasets/nuscenes/analyse_boxes.py:6-26 (no pillar height limit): the float32 points are promoted to
    float64 by the float64 z range, scaled, and converted to int32 by TRUNCATION (so (-1, 0) lands in cell 0 and counts as inside);
    returns the cell of every point and the in-range mask"""
    r = np.append(np.array(bev_range_m, np.float32), np.array(1000.0))
    gs = np.append(np.array(grid).astype(np.int32), np.array(1))
    p = pcl.numpy() if torch.is_tensor(pcl) else pcl
    v = (((p[:, :3] + 0.5 * r) / r) * gs).astype(np.int32)
    inside = np.all((0 <= v) & (v < gs), axis=-1)
    return v[:, :2], inside


def augment(sample_pcl, sample_coors, sample_flow, db, bev_range_m, grid, box_cfg, need_flow=True, centers=None):
    """One augmented sample; `db` = {"points": [M] list of [n_i, 4] float32, "dims": [M, 3] float32, "pos_z": [M] float32}.
    Returns a dict of numpy arrays: boxes (pos float64, dims, rot, velo), the cloud with the pasted points appended (before and
    after the in-range filter), pillar coordinates, flows, and the intermediate free mask."""
    G = np.array(grid).astype(np.int32)
    num = np.random.randint(low=1, high=box_cfg["max_num_objs"] + 1)
    ppm = (G / np.array(bev_range_m, np.float32)).astype(np.float32)
    size_pillar = 1 / ppm
    radius = max(3, int(box_cfg.get("min_obj_center_dist_from_occupied_pillars_m", 2.0) / size_pillar.mean()))
    free = free_location_mask(sample_coors, G, radius)
    centers = bev_center_coords(bev_range_m, grid) if centers is None else centers
    allowed = centers[free]
    loc_idx = np.random.choice(np.arange(allowed.shape[0]), size=num, replace=False)
    xy = torch.from_numpy(allowed[loc_idx][:, :2])
    xy += (0.5 - torch.rand_like(xy)) * torch.from_numpy(size_pillar)
    obj = np.random.choice(np.arange(len(db["points"])), size=num, replace=True)
    dims = db["dims"][obj]
    z_old = torch.from_numpy(db["pos_z"][obj][:, None])
    z_new = 0.5 * (torch.rand((num, 1)) - 0.5) + z_old
    rot = 2 * np.pi * (torch.rand((num, 1)) - 0.5)
    pos = torch.cat([xy, z_new], dim=-1)
    s, c = torch.sin(rot[:, 0]).double().numpy(), torch.cos(rot[:, 0]).double().numpy()
    T = np.tile(np.eye(4), (num, 1, 1))
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = c, -s, s, c
    T[:, 0, 3], T[:, 1, 3] = pos[:, 0].double().numpy(), pos[:, 1].double().numpy()
    velo = np.zeros((num, 1), np.float32)
    extra_pcl, extra_flow = [], []
    vmin, vmax = box_cfg["min_artificial_obj_velo"], box_cfg["max_artificial_obj_velo"]
    for i, o in enumerate(obj):
        p = np.copy(db["points"][o])
        if box_cfg["max_points_dropout"] != 0.0:
            n = p.shape[0]
            keep = max(1, int(n * (1.0 - np.random.rand() * box_cfg["max_points_dropout"])))
            p = p[np.random.choice(np.arange(0, n, 1, dtype=int), keep, replace=False)]
        fx = 1 if np.random.rand() < 0.5 else -1
        fy = 1 if np.random.rand() < 0.5 else -1
        sx = 1.0 - box_cfg["max_scale_delta"] * (2 * np.random.rand() - 1.0)
        sy = 1.0 - box_cfg["max_scale_delta"] * (2 * np.random.rand() - 1.0)
        sz = 1.0 - box_cfg["max_scale_delta"] * (2 * np.random.rand() - 1.0)
        F = np.eye(4)
        F[0, 0], F[1, 1], F[2, 2] = fx * sx, fy * sy, sz
        hom = np.concatenate([p[:, :3], np.ones_like(p[:, :1])], -1)
        ps = np.einsum("ij,nj->ni", T[i] @ F, hom)[:, :3]
        fl = vmin + np.random.rand(*ps.shape) * (vmax - vmin)
        velo[i] = np.mean(np.linalg.norm(fl[:, :3], axis=-1, keepdims=True), axis=0)
        extra_pcl.append(np.concatenate([ps, p[:, [-1]]], -1))
        extra_flow.append(fl)
    extra_pcl = np.concatenate(extra_pcl, 0).astype(np.float32)
    cloud = np.concatenate([sample_pcl, extra_pcl], 0)
    coors, in_range = voxelize_sample(cloud, bev_range_m, grid)
    out = {"free_mask": free, "box_pos": pos.numpy(), "box_dims": dims, "box_rot": rot.numpy(), "box_velo": velo,
           "extra_pcl": extra_pcl, "pcl_all": cloud, "pcl": cloud[in_range], "pillar_coors": coors[in_range], "in_range": in_range}
    if need_flow:
        flow = np.concatenate([sample_flow, np.concatenate(extra_flow, 0).astype(np.float32)], 0)
        out["flow"] = flow[in_range]
    return out
