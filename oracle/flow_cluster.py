"""TEST INFRASTRUCTURE ONLY: torch-CPU restatement of rows D1 and D3.  Pinned by
tests/golden/flow_cluster_reference.npz generated from the reference's own python
(tests/golden/make_flow_cluster_golden.py)."""
import torch


def scatter_mean_2d(valid, batch_idx, rows, cols, values, B, H, W):
    """liso/utils/torch_differentiable_forward_scatter.py:22-87: scatter_add of values + counts, divide where count > 1"""
    C = values.shape[-1]
    v = values[valid].double()
    lin = ((batch_idx[valid] * H + rows[valid].long()) * W + cols[valid].long())
    tgt = torch.zeros(B * H * W, C, dtype=torch.float64)
    tgt.index_add_(0, lin, v)
    cnt = torch.zeros(B * H * W, dtype=torch.int64)
    cnt.index_add_(0, lin, torch.ones_like(lin))
    out = torch.where(cnt[:, None] > 1, tgt / cnt[:, None].clamp(min=1), tgt)
    return out.float().view(B, H, W, C)


def bev_dynamic_flow(pcl_is_valid, pcl, pillar_coors, point_flow, odom_ta_tb, target_shape):
    """liso/utils/bev_flow_utils.py:6-77 -> (dynamicness[B,H,W,1], nonrigid_flow[B,H,W,3])"""
    homog = torch.cat([pcl[..., :3], torch.ones_like(pcl[..., :1])], -1)
    homog = torch.where(pcl_is_valid[..., None], homog, torch.zeros(()))
    flow = torch.where(pcl_is_valid[..., None], point_flow, torch.zeros(()))
    M = torch.linalg.inv(odom_ta_tb.double()) - torch.eye(4, dtype=torch.float64)[None]
    stat = torch.einsum("bij,bnj->bni", M, homog.double())[..., :3].float()
    stat = torch.where(pcl_is_valid[..., None], stat, torch.zeros(()))
    nonrigid = flow - stat
    length = torch.linalg.norm(nonrigid, dim=-1, keepdim=True)
    B, N = pcl_is_valid.shape
    bidx = torch.arange(B)[:, None].repeat(1, N)
    H, W = int(target_shape[0]), int(target_shape[1])
    return (scatter_mean_2d(pcl_is_valid, bidx, pillar_coors[..., 0], pillar_coors[..., 1], length, B, H, W),
            scatter_mean_2d(pcl_is_valid, bidx, pillar_coors[..., 0], pillar_coors[..., 1], nonrigid, B, H, W))


def fit_box_z(pcl, pos, dims, rot, box_height=1000.0):
    """flow_cluster_detector.py:339-384"""
    K = pos.shape[0]
    c, s = torch.cos(rot.double()), torch.sin(rot.double())
    bz = pos[:, 2].double() if pos.shape[-1] == 3 else torch.zeros(K, dtype=torch.float64)
    dx = pcl[:, None, 0].double() - pos[None, :, 0].double()
    dy = pcl[:, None, 1].double() - pos[None, :, 1].double()
    lx, ly = (c * dx + s * dy).float(), (-s * dx + c * dy).float()
    lz = (pcl[:, None, 2].double() - bz[None]).float()
    d = dims if dims.shape[-1] == 3 else torch.cat([dims, box_height * torch.ones_like(dims[:, :1])], -1)
    inside = (lx.abs() < 0.5 * d[None, :, 0]) & (ly.abs() < 0.5 * d[None, :, 1]) & (lz.abs() < 0.5 * d[None, :, 2])
    zmax = torch.where(inside, lz, torch.tensor(-box_height)).max(dim=0).values
    zmin = torch.where(inside, lz, torch.tensor(box_height)).min(dim=0)
    h = torch.clip(zmax - zmin.values, min=1.0, max=2.0)
    return inside.sum(0), pcl[:, 2][zmin.indices] + 0.5 * h, h


# ---- D2: clustering block (flow_cluster_detector.py:151-189) ---------------------------------------------------------
# Both libraries are third-party arithmetic: scikit-learn (pinned 0.24.2 by the reference; 1.7.2 in this
# image -- DBSCAN is deterministic given the point order, the labelling rule has not changed) is CALLED here exactly as
# the reference calls it; scikit-image (pinned 0.19.2) is absent from the image, so regionprops' four properties are
# restated from its published formulas (skimage/measure/_regionprops.py, _moments.py: inertia_tensor,
# inertia_tensor_eigvals, orientation, axis_major_length / axis_minor_length).  Pinned (round 2) by
# tests/golden/regionprops_reference.npz, which scikit-image 0.18.3 itself produced (tests/golden/make_regionprops_golden.py, run
# with the build container's /opt/conda/bin/python3.9): tests/test_regionprops_reference.py; analytic shapes in
# tests/test_oracle_flow_cluster.py.
def dbscan_bev_labels(valid_mask, bev_nonrigid_flow, grid_pts_xy, eps=1.0, min_samples=5, flow_similarity_importance=2.0):
    """one sample: valid_mask [G,G] bool numpy, bev_nonrigid_flow [G,G,3] float32 numpy, grid_pts_xy [G,G,2] float32
    -> label image int64 [G,G] (0 = background / noise), exactly lines :151-172 of the reference"""
    import numpy as np
    from sklearn.cluster import DBSCAN

    bev_labels = np.zeros(valid_mask.shape, dtype=np.int64)
    if np.count_nonzero(valid_mask) <= 1:
        return bev_labels
    dynamic_coors = grid_pts_xy[valid_mask]
    dynamic_flow = flow_similarity_importance * bev_nonrigid_flow[valid_mask]
    cluster_coords = np.concatenate([dynamic_coors, dynamic_flow], axis=-1)
    db = DBSCAN(eps=eps, min_samples=min_samples, metric="euclidean", algorithm="auto", n_jobs=1).fit(cluster_coords)
    labels = np.where(db.labels_ >= 0, db.labels_ + 1, 0)
    rows, cols = np.nonzero(valid_mask)
    bev_labels[rows, cols] = labels
    return bev_labels


def regionprops_restated(label_img):
    """skimage.measure.regionprops(label_img) -> float64 [K,5] rows (centroid_row, centroid_col, orientation,
    axis_major_length, axis_minor_length) for labels 1..K in ascending order (labels without pixels are skipped by
    skimage; DBSCAN labels are dense so none are)."""
    import math

    import numpy as np

    out = []
    for lab in range(1, int(label_img.max()) + 1):
        rr, cc = np.nonzero(label_img == lab)
        if rr.size == 0:
            continue
        r0, c0 = rr.mean(), cc.mean()
        mu00 = float(rr.size)
        mu20 = ((rr - r0) ** 2).sum() / mu00  # second central moment along rows (axis 0)
        mu02 = ((cc - c0) ** 2).sum() / mu00
        mu11 = ((rr - r0) * (cc - c0)).sum() / mu00
        # _moments.inertia_tensor: [[mu02, -mu11], [-mu11, mu20]] / mu00
        a, b, c = mu02, -mu11, mu20
        if a - c == 0:  # _regionprops.orientation
            orientation = -math.pi / 4.0 if b < 0 else math.pi / 4.0
        else:
            orientation = 0.5 * math.atan2(-2 * b, c - a)
        ev = np.clip(np.sort(np.linalg.eigvalsh(np.array([[a, b], [b, c]])))[::-1], 0.0, None)  # inertia_tensor_eigvals
        out.append([r0, c0, orientation, 4.0 * math.sqrt(ev[0]), 4.0 * math.sqrt(ev[1])])
    return np.array(out, dtype=np.float64).reshape(-1, 5)


def flow_cluster_detector_forward(pcl, pcl_is_valid, pcl_w_ground, pillar_coors, point_flow, odom_ta_tb, time_delta_s,
                                  grid_pts_xy, pix_per_m, *, min_num_pts_per_box=10, max_box_len_m=7.0, aspect_ratio_max=4.0,
                                  min_box_area_m2=0.35, min_box_volume_m3=0.5, slope=15.0, buffer=0.25):
    """FlowClusterDetector.forward (flow_cluster_detector.py:87-336) restated on the CPU from the oracle pieces.
    All tensors CPU torch; grid_pts_xy [G,G,2] float32 numpy (pcl_bev_center_coords_homog[..., :2]), pix_per_m [2] float32.
    -> dict(pos [B,S,3], dims [B,S,3], rot [B,S,1], velo [B,S,1], valid [B,S], labels [B,G,G])"""
    import numpy as np

    from . import kabsch as OK

    B = pcl.shape[0]
    G = grid_pts_xy.shape[0]
    dyn, nrf = bev_dynamic_flow(pcl_is_valid, pcl, pillar_coors, point_flow, odom_ta_tb, (G, G))
    mask = (dyn[..., 0] > (time_delta_s * 1.0)[:, None, None]).numpy()
    per_sample, label_imgs = [], []
    for b in range(B):
        lab = dbscan_bev_labels(mask[b], nrf[b].numpy(), grid_pts_xy)
        label_imgs.append(lab)
        props = regionprops_restated(lab)
        K = props.shape[0]
        pix = np.clip(props[:, 0:2].astype(int), 0, G - 1)
        pos2 = torch.from_numpy(grid_pts_xy[pix[:, 0], pix[:, 1]]).reshape(K, 2)
        rot = torch.from_numpy(props[:, 2:3])
        dims2 = torch.from_numpy(props[:, 3:5]) * 1.0 / torch.from_numpy(pix_per_m)
        if K > 0:
            n, z, h = fit_box_z(pcl_w_ground[b][:, :3], pos2, dims2.float(), rot[:, 0].float(), box_height=1000.0)
        else:
            n, z, h = torch.zeros(0, dtype=torch.int64), torch.zeros(0), torch.zeros(0)
        ok = n >= min_num_pts_per_box
        ok &= dims2[:, 0] / torch.max(dims2[:, 1], 0.001 * torch.ones_like(dims2[:, 1])) <= aspect_ratio_max
        ok &= dims2[:, 0] <= max_box_len_m
        ok &= torch.prod(dims2, dim=-1) > min_box_area_m2
        dims3 = torch.cat([dims2, h[:, None].double()], dim=-1)
        pos3 = torch.cat([pos2, z[:, None]], dim=-1)
        ok &= torch.prod(dims3, dim=-1) > min_box_volume_m3
        per_sample.append((pos3[ok], dims3[ok], rot[ok]))
    S = max(p[0].shape[0] for p in per_sample)
    pos = torch.zeros(B, S, 3)
    dims = torch.zeros(B, S, 3, dtype=torch.float64)
    rot = torch.zeros(B, S, 1, dtype=torch.float64)
    valid = torch.zeros(B, S, dtype=torch.bool)
    for b, (p, d, r) in enumerate(per_sample):
        k = p.shape[0]
        pos[b, :k], dims[b, :k], rot[b, :k], valid[b, :k] = p, d, r, True
    velo = torch.zeros(B, S, 1, dtype=torch.float64)
    if S > 0:
        T, _, _ = OK.kabsch_trafos(pos, dims.float(), rot.float(), pcl[..., :3], pcl_is_valid, point_flow, slope=slope, buffer=buffer)
        fg, bg = T[:, :-1], T[:, -1:]
        c, s = torch.cos(rot[..., 0].double()), torch.sin(rot[..., 0].double())
        Tb = torch.zeros(B, S, 4, 4, dtype=torch.float64)  # shape_utils.py:271-319: translation * yaw
        Tb[..., 0, 0], Tb[..., 0, 1], Tb[..., 1, 0], Tb[..., 1, 1] = c, -s, s, c
        Tb[..., 0, 3], Tb[..., 1, 3], Tb[..., 2, 3] = pos[..., 0].double(), pos[..., 1].double(), pos[..., 2].double()
        Tb[..., 2, 2] = Tb[..., 3, 3] = 1.0
        M = torch.linalg.inv(Tb) @ torch.linalg.inv(bg) @ (fg @ Tb)  # shape_utils.py:583-605
        tr = M[..., :3, 3]
        rot = rot + torch.atan2(tr[..., [1]], tr[..., [0]])
        velo[..., 0] = torch.linalg.norm(tr, dim=-1)
    return dict(pos=pos, dims=dims, rot=rot, velo=velo, valid=valid, labels=np.stack(label_imgs))
