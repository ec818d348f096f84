"""TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product may not).

CPU restatement (numpy, float64) of the tracker's local box refinement:
  liso/box_fitting/box_fitting.py:93-141,242-258   closeness_rectangle / fit_2d_box_modest("closeness_to_edge")
  liso/tracker/tracking.py:239-260                 set_box_size_keep_closest_point_constant
  liso/tracker/tracking.py:2004-2133               perform_local_box_refinement
Pinned by tests/golden/box_refinement_reference.npz (written by the reference's own functions, tests/golden/
make_box_refinement_golden.py)."""
import numpy as np

ANGLES_DEG = np.arange(0, 90 + 5.0, 5.0)  # the 19 candidate headings, box_fitting.py:96


def _extents(xy, angle):
    c, s = np.cos(angle), np.sin(angle)
    px, py = xy[:, 0] * c + xy[:, 1] * s, -xy[:, 0] * s + xy[:, 1] * c
    return px, py, px.min(), px.max(), py.min(), py.max()


def closeness_fit(xy, d0=1e-2):
    """-> (centre xy, length, width, yaw): the heading whose bounding rectangle has the points closest to its edges (sum of
    1 / max(distance to the nearest edge, d0); first maximum wins), turned by 90 deg if that makes the x side the long one"""
    xy = np.asarray(xy, np.float64)
    best, angle = -np.inf, 0.0
    for deg in ANGLES_DEG:
        a = deg / 180.0 * np.pi
        px, py, x0, x1, y0, y1 = _extents(xy, a)
        beta = (1.0 / np.maximum(np.minimum(np.minimum(px - x0, x1 - px), np.minimum(py - y0, y1 - py)), d0)).sum()
        if beta > best:
            best, angle = beta, a
    _, _, x0, x1, y0, y1 = _extents(xy, angle)
    if (x1 - x0) < (y1 - y0):
        angle = angle + np.pi / 2
        _, _, x0, x1, y0, y1 = _extents(xy, angle)
    c, s = np.cos(angle), np.sin(angle)
    comp = np.array([[c, s], [-s, c]])
    corners = np.array([[x1, y0], [x0, y0], [x0, y1], [x1, y1]]) @ comp
    return (corners[0] + corners[2]) / 2, np.linalg.norm(corners[0] - corners[1]), np.linalg.norm(corners[0] - corners[-1]), angle


def points_in_bloated_footprint(cloud_xyz, pos, dims, yaw, bloat):
    """tracking.py:2044-2056: fp64 transform into the box frame, compared as float32 against 0.5 * bloat * dims (x and y only)"""
    c, s = np.cos(np.float64(yaw)), np.sin(np.float64(yaw))
    d = cloud_xyz[:, :2].astype(np.float64) - np.asarray(pos[:2], np.float64)
    bx, by = c * d[:, 0] + s * d[:, 1], -s * d[:, 0] + c * d[:, 1]
    half = np.asarray(dims[:2], np.float32) * np.float32(0.5 * bloat)
    return (np.abs(bx.astype(np.float32)) < half[0]) & (np.abs(by.astype(np.float32)) < half[1])


def set_box_size_keep_closest_point_constant(pos, dims, yaw, new_dims):
    """tracking.py:239-260: resize every box about its bottom corner closest to the sensor; float64 pose, float32 boxes"""
    pos, dims, new_dims = np.asarray(pos, np.float32), np.asarray(dims, np.float32), np.asarray(new_dims, np.float32)
    out = np.empty_like(pos)
    unit = 0.5 * np.array([(1.0, -1.0, -1.0), (1.0, 1.0, -1.0), (-1.0, -1.0, -1.0), (-1.0, 1.0, -1.0)])  # bottom corners 0, 1, 4, 5
    for i in range(pos.shape[0]):
        c, s = np.cos(np.float64(yaw[i, 0])), np.sin(np.float64(yaw[i, 0]))
        local = (unit.astype(np.float32) * dims[i]).astype(np.float64)
        corners = np.stack([c * local[:, 0] - s * local[:, 1] + pos[i, 0], s * local[:, 0] + c * local[:, 1] + pos[i, 1],
                            local[:, 2] + pos[i, 2]], -1)
        k = int(np.argmin(np.linalg.norm(corners[:, :2], axis=-1)))
        closest = corners[k]
        out[i] = (closest + (new_dims / dims[i]).astype(np.float64) * (pos[i].astype(np.float64) - closest)).astype(np.float32)
    return out, np.ones_like(dims) * new_dims


def perform_local_box_refinement(clouds, pos, dims, rot, track_age, start_time_idx, fit_rot, fit_pos, bloat, dims_quantile):
    pos, dims, rot = np.array(pos, np.float32), np.array(dims, np.float32), np.array(rot, np.float32)
    new_dims = np.quantile(dims.astype(np.float64), dims_quantile, axis=0).astype(np.float32)
    if fit_rot or fit_pos:
        for t in range(track_age):
            cloud = clouds[start_time_idx + t]
            inside = points_in_bloated_footprint(cloud[:, :3], pos[t], dims[t], rot[t, 0], bloat)
            if inside.any():
                center, _, _, yaw = closeness_fit(cloud[inside, :2].astype(np.float64))
                if fit_rot:
                    rot[t, 0] = rot[t, 0] + np.float32(yaw - float(rot[t, 0]))
                if fit_pos:
                    pos[t, :2] = center.astype(np.float32)
    pos, dims = set_box_size_keep_closest_point_constant(pos, dims, rot, new_dims)
    return pos, dims, rot
