"""Bird's-eye rectangle fits of a point cluster.  Mirror of liso/box_fitting/box_fitting.py:10-258 (same function names, arguments
and return values: (4 corners [4,2], yaw, area) per fit, `fit_2d_box_modest` -> (centre, length, width, yaw)).

Only "closeness_to_edge" is on a call path of the reference (tracking.py:2064 -- the per-track box refinement, which this build
runs for all boxes of a frame in one launch: liso_fit_boxes_closeness_f32, liso_amd/tracker/tracking.py::fit_boxes_to_points); the
other three criteria are selectable there but never selected.  They are host functions here as there, written over all candidate
angles at once instead of a Python loop per angle."""
import numpy as np


def _axes(angle):
    """rows = the rectangle's axes at `angle` (box_fitting.py:99-101); angle may be an array -> [A,2,2]"""
    c, s = np.cos(angle), np.sin(angle)
    return np.stack([np.stack([c, s], -1), np.stack([-s, c], -1)], -2)


def _rectangle_at(points, angle):
    """bounding rectangle of the points in the frame at `angle`, long side first (box_fitting.py:114-141) -> corners, angle, area"""
    for a in (angle, angle + np.pi / 2):
        axes = _axes(a)
        proj = points @ axes.T
        lo, hi = proj.min(axis=0), proj.max(axis=0)
        if a is angle and (hi[0] - lo[0]) >= (hi[1] - lo[1]):
            break
    corners = np.array([[hi[0], lo[1]], [lo[0], lo[1]], [lo[0], hi[1]], [hi[0], hi[1]]]) @ axes
    return corners, a, (hi[0] - lo[0]) * (hi[1] - lo[1])


def _edge_distances(points, angles):
    """per candidate angle: distance of every point to the nearer bounding edge along either axis -> Dx, Dy [A,N]"""
    proj = np.einsum("nk,ajk->ajn", points, _axes(angles))  # [A,2,N]
    lo, hi = proj.min(axis=2, keepdims=True), proj.max(axis=2, keepdims=True)
    d = np.minimum(proj - lo, hi - proj)
    return d[:, 0], d[:, 1]


def closeness_rectangle(cluster_ptc, delta=5.0, d0=1e-2):
    """reference :93-141 -- the angle in [0, 90] deg (step `delta`) that maximises sum 1 / max(d_edge, d0); first maximum wins"""
    angles = np.arange(0, 90 + delta, delta) / 180.0 * np.pi
    dx, dy = _edge_distances(cluster_ptc, angles)
    beta = (1 / np.maximum(np.minimum(dx, dy), d0)).sum(axis=1)
    best = 0.0 if not np.any(beta > -np.inf) else angles[int(np.argmax(beta))]
    return _rectangle_at(cluster_ptc, best)


def variance_rectangle(cluster_ptc, delta=0.1):
    """reference :144-199 -- the angle that minimises the variance of the edge distances of the points assigned to each axis"""
    angles = np.arange(0, 90 + delta, delta) / 180.0 * np.pi
    dx, dy = _edge_distances(cluster_ptc, angles)
    score = np.zeros(len(angles))
    for mask, d in ((dx < dy, dx), (dy < dx, dy)):
        n = mask.sum(axis=1)
        mean = np.where(n > 0, (d * mask).sum(axis=1) / np.maximum(n, 1), 0.0)
        var = (((d - mean[:, None]) ** 2) * mask).sum(axis=1) / np.maximum(n, 1)
        score += np.where(n > 0, -var, 0.0)
    return _rectangle_at(cluster_ptc, angles[int(np.argmax(score))])


def PCA_rectangle(cluster_ptc):
    """reference :70-87 -- axes = principal components of the cluster (scikit-learn's PCA: descending variance, every component's
    sign such that its largest entry is positive)"""
    x = np.asarray(cluster_ptc, dtype=np.float64)
    centred = x - x.mean(axis=0)
    _, _, vt = np.linalg.svd(centred, full_matrices=False)
    big = np.argmax(np.abs(vt), axis=1)
    vt = vt * np.sign(vt[np.arange(vt.shape[0]), big])[:, None]
    proj = cluster_ptc @ vt.T
    lo, hi = proj.min(axis=0), proj.max(axis=0)
    corners = np.array([[hi[0], lo[1]], [lo[0], lo[1]], [lo[0], hi[1]], [hi[0], hi[1]]]) @ vt
    return corners, np.arctan2(vt[0, 1], vt[0, 0]), (hi[0] - lo[0]) * (hi[1] - lo[1])


def minimum_bounding_rectangle(points):
    """reference :10-67 -- smallest-area rectangle with a side along an edge of the convex hull (the hull's closing edge is not a
    candidate there, and so not here: hull vertices in qhull's order)"""
    from scipy.spatial import ConvexHull

    hull = points[ConvexHull(points).vertices]
    edges = hull[1:] - hull[:-1]
    angles = np.unique(np.abs(np.mod(np.arctan2(edges[:, 1], edges[:, 0]), np.pi / 2.0)))
    h = np.pi / 2.0
    rot = np.vstack([np.cos(angles), np.cos(angles - h), np.cos(angles + h), np.cos(angles)]).T.reshape((-1, 2, 2))
    turned = np.dot(rot, hull.T)  # [A,2,H]
    lo, hi = np.nanmin(turned, axis=2), np.nanmax(turned, axis=2)
    areas = (hi[:, 0] - lo[:, 0]) * (hi[:, 1] - lo[:, 1])
    b = int(np.argmin(areas))
    x1, x2, y1, y2 = hi[b, 0], lo[b, 0], hi[b, 1], lo[b, 1]
    corners = np.array([np.dot([x1, y2], rot[b]), np.dot([x2, y2], rot[b]), np.dot([x2, y1], rot[b]), np.dot([x1, y1], rot[b])])
    return corners, angles[b], areas[b]


_METHODS = {"min_zx_area_fit": minimum_bounding_rectangle, "PCA": PCA_rectangle, "variance_to_edge": variance_rectangle,
            "closeness_to_edge": closeness_rectangle}


def fit_2d_box_modest(ptc, fit_method):
    """reference :242-258 -- (centre xy, length, width, yaw) of the fitted rectangle of the points' x, y"""
    assert ptc.shape[-1] == 3, ptc.shape
    if fit_method not in _METHODS:
        raise NotImplementedError(fit_method)
    corners, ry, _ = _METHODS[fit_method](ptc[:, [0, 1]])
    return (corners[0] + corners[2]) / 2, np.linalg.norm(corners[0] - corners[1]), np.linalg.norm(corners[0] - corners[-1]), ry
