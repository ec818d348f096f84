"""TEST INFRASTRUCTURE ONLY: CPU restatement of the pillar path.

numpy for the integer voxelisation, fp32 torch-CPU for the PillarFeatureNet arithmetic (the "plain PyTorch fp32
reference" of a floating-point kernel; autograd through it yields the reference gradients).  Every function cites
the reference lines it follows (paths relative to /root/reference).

Pinning: tests/golden/pillars_*.npz are produced by tests/golden/make_pillars_golden.py from the reference's own
python (mmdet3d voxel_generator / PillarFeatureNet / PFNLayer / PointPillarsScatter imported from
/root/reference with registry/decorator-only stubs for the absent mmcv/numba) and checked in
tests/test_oracle_pillars.py.  mmcv.ops.Voxelization itself (CUDA, un-vendored mmcv-full 1.7.1,
docker/Dockerfile.base:68) cannot run here: its canonical semantics are taken to be the in-tree CPU twin
(first-come-first-served), see DESIGN.md.
"""
import numpy as np
import torch
import torch.nn.functional as F


def pillar_geometry(bev_range_m, img_grid_size, z_cut):
    """liso/networks/pcl_to_feature_grid/pcl_to_feature_grid.py:14-23 -> (pc_range[6], voxel_size[3]) float64"""
    half = np.append(np.array(bev_range_m, dtype=np.float64) / 2.0, z_cut)
    pc_range = np.concatenate([-half, half], axis=0)
    voxel_size = np.append(np.array(bev_range_m, dtype=np.float64) / np.array(img_grid_size, dtype=np.float64), 2 * z_cut)
    return pc_range, voxel_size


def voxelize_hard(points, voxel_size, pc_range, max_points=20, max_voxels=40000):
    """mmdetection3d/mmdet3d/core/voxel/voxel_generator.py:76-134 + :211-280 (reverse_index=True):
    first-come-first-served hard voxelisation in point order.  Returns
      voxels[P,max_points,C] float32, coors[P,3] int32 in (z,y,x) order, num_points[P] int32, point_idx[P,max_points]
    (point_idx is extra: which input point sits in each slot, -1 = padding).  fp32 arithmetic like the reference
    (voxel_size / coors_range are cast to points.dtype, :105-108)."""
    points = np.asarray(points)
    dt = points.dtype if points.dtype in (np.float32, np.float64) else np.float32
    pts = np.ascontiguousarray(points, dtype=dt)
    vs = np.asarray(voxel_size, dtype=dt)
    rng = np.asarray(pc_range, dtype=dt)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int32)  # :242-245
    c = np.floor((pts[:, :3] - rng[:3]) / vs)                   # :257
    ok = np.all((c >= 0) & (c < grid), axis=1)                  # :258-260 (NaN compares false -> kept by the
    ok &= np.all(np.isfinite(pts[:, :3]), axis=1)               #  reference; we define NaN points as dropped)
    idx = np.nonzero(ok)[0]
    ci = c[idx].astype(np.int64)
    lin = (ci[:, 2] * grid[1] + ci[:, 1]) * grid[0] + ci[:, 0]  # (z,y,x) linearised
    uniq, first, inv = np.unique(lin, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                    # voxel ordinal = order of first appearance, :265-272
    rank_of_uniq = np.empty_like(order)
    rank_of_uniq[order] = np.arange(len(order))
    vox_of_pt = rank_of_uniq[inv]
    keep = vox_of_pt < max_voxels                               # :267-268 voxels beyond max_voxels are skipped
    idx, vox_of_pt = idx[keep], vox_of_pt[keep]
    P = int(min(len(uniq), max_voxels))
    # slot of each point inside its voxel = running count in point order, :273-277
    srt = np.argsort(vox_of_pt, kind="stable")
    v_sorted = vox_of_pt[srt]
    start = np.searchsorted(v_sorted, np.arange(P), side="left")
    slot = np.arange(len(srt)) - start[v_sorted]
    sel = slot < max_points
    point_idx = -np.ones((P, max_points), np.int64)
    point_idx[v_sorted[sel], slot[sel]] = idx[srt][sel]
    num = np.minimum(np.bincount(vox_of_pt, minlength=P), max_points).astype(np.int32)
    voxels = np.zeros((P, max_points, pts.shape[1]), dt)
    valid = point_idx >= 0
    voxels[valid] = pts[point_idx[valid]]
    first_pt = idx[srt][start] if P else np.zeros((0,), np.int64)
    cz = np.floor((pts[first_pt, :3] - rng[:3]) / vs).astype(np.int32) if P else np.zeros((0, 3), np.int32)
    coors = cz[:, ::-1].copy()                                  # reverse_index: (z,y,x), :262
    return voxels, coors, num, point_idx


def voxelize_batch(pcls, voxel_size, pc_range, max_points=20, max_voxels=40000):
    """pcl_to_feature_grid.py:58-84: per-sample voxelise, swap to (z, x, y) (:73), prepend batch index (:79-83)."""
    V, N, Cc, PI = [], [], [], []
    off = 0
    for b, p in enumerate(pcls):
        p = np.asarray(p, np.float32)
        v, c, n, pi = voxelize_hard(p, voxel_size, pc_range, max_points, max_voxels)
        c = c[:, [0, 2, 1]]
        V.append(v)
        N.append(n)
        Cc.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], axis=1))
        PI.append(np.where(pi >= 0, pi + off, -1))
        off += len(p)
    return np.concatenate(V), np.concatenate(N), np.concatenate(Cc), np.concatenate(PI)


def pfn_decorate(voxels, num_points, coors, voxel_size, pc_range):
    """pillar_encoder.py:109-152 with legacy=True: the in-place f_center update aliases features[:, :, :3]."""
    features = voxels.clone()
    vx, vy, vz = (float(v) for v in voxel_size)
    x_off, y_off, z_off = vx / 2 + float(pc_range[0]), vy / 2 + float(pc_range[1]), vz / 2 + float(pc_range[2])  # :86-91
    points_mean = features[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(features).view(-1, 1, 1)
    f_cluster = features[:, :, :3] - points_mean
    f_center = features[:, :, :3]  # a view: the updates below also rewrite `features` (:129)
    f_center[:, :, 0] = f_center[:, :, 0] - (coors[:, 3].type_as(features).unsqueeze(1) * vx + x_off)
    f_center[:, :, 1] = f_center[:, :, 1] - (coors[:, 2].type_as(features).unsqueeze(1) * vy + y_off)
    f_center[:, :, 2] = f_center[:, :, 2] - (coors[:, 1].type_as(features).unsqueeze(1) * vz + z_off)
    feats = torch.cat([features, f_cluster, f_center], dim=-1)
    mask = torch.arange(voxels.shape[1])[None, :] < num_points[:, None]  # utils.py:9-29
    return feats * mask.unsqueeze(-1).type_as(feats)


def pfn_layer(feats, weight, gamma, beta, running_mean, running_var, training, momentum=0.01, eps=1e-3):
    """utils.py:162-176 (mode='max', last layer): Linear -> BatchNorm1d over [P,64,20] -> ReLU -> max over slots."""
    x = F.linear(feats, weight)
    x = F.batch_norm(x.permute(0, 2, 1).contiguous(), running_mean, running_var, gamma, beta, training, momentum, eps)
    x = F.relu(x.permute(0, 2, 1).contiguous())
    return torch.max(x, dim=1, keepdim=True)[0].squeeze(1)


def scatter(voxel_features, coors, batch_size, ny, nx):
    """pillar_scatter.py:62-102: canvas[:, coors[:,2] * nx + coors[:,3]] = features, per sample -> [B,C,ny,nx]"""
    C = voxel_features.shape[1]
    out = []
    for b in range(batch_size):
        canvas = torch.zeros(C, nx * ny, dtype=voxel_features.dtype)
        m = coors[:, 0] == b
        ind = (coors[m, 2] * nx + coors[m, 3]).long()
        canvas[:, ind] = voxel_features[m].t()
        out.append(canvas)
    return torch.stack(out, 0).view(batch_size, C, ny, nx)


def pillar_forward(pcls, weight, gamma, beta, running_mean, running_var, training, bev_range_m, img_grid_size, z_cut,
                   max_points=20, max_voxels=40000, dtype=torch.float32):
    """pcl_to_feature_grid.py:86-107 end to end -> (bev[B,64,H,W], occupancy[B,1,H,W], (voxels,num,coors,point_idx))"""
    pc_range, voxel_size = pillar_geometry(bev_range_m, img_grid_size, z_cut)
    v, n, c, pi = voxelize_batch(pcls, voxel_size, pc_range, max_points, max_voxels)
    vt, nt, ct = torch.from_numpy(v).to(dtype), torch.from_numpy(n), torch.from_numpy(c)
    feats = pfn_decorate(vt, nt, ct, voxel_size, pc_range)
    vf = pfn_layer(feats, weight, gamma, beta, running_mean, running_var, training)
    B = len(pcls)
    bev = scatter(vf, ct, B, img_grid_size[0], img_grid_size[1])
    occ = scatter(torch.ones_like(vf[:, [0]]), ct, B, img_grid_size[0], img_grid_size[1])
    return bev, occ, (v, n, c, pi)


def synthetic_cloud(n, seed, bev_range_m=100.0, n_channels=4, dense_clusters=True):
    """SURVEY.md 8d config 1 generator (xy~U(-R/2,R/2), z~U(-2,1), i~U(0,1)) plus a few dense blobs so that some
    pillars exceed 20 points and some points fall outside the range."""
    r = np.random.default_rng(seed)
    p = np.zeros((n, n_channels), np.float32)
    p[:, 0:2] = r.uniform(-bev_range_m * 0.52, bev_range_m * 0.52, (n, 2))
    p[:, 2] = r.uniform(-2, 1, n)
    if n_channels > 3:
        p[:, 3:] = r.uniform(0, 1, (n, n_channels - 3))
    if dense_clusters and n >= 200:
        k = n // 5
        ctr = r.uniform(-bev_range_m * 0.3, bev_range_m * 0.3, (4, 2))
        which = r.integers(0, 4, k)
        p[:k, 0:2] = ctr[which] + r.normal(0, 0.15, (k, 2))
        r.shuffle(p, axis=0)
    return p
