"""Generates tests/golden/dbscan_reference.npz with the libraries the reference itself calls, in the versions the build container's
second interpreter carries: scikit-learn 0.24.2 (exactly the reference's pin) and scikit-image 0.18.3 --
    sklearn.cluster.DBSCAN(eps=1.0, min_samples=5, metric="euclidean", algorithm="auto", n_jobs=1).fit(...)  then
    skimage.measure.regionprops(label image)
as liso/networks/flow_cluster_detector/flow_cluster_detector.py:151-189 strings them together: features (x, y, 2 fx, 2 fy, 2 fz)
of the dynamic pillars, labels + 1 with noise -> 0 written into the BEV grid, one property row per cluster.
Run in the build container only:  /opt/conda/bin/python3.9 tests/golden/make_dbscan_golden.py
"""
import os

import numpy as np
import skimage
import sklearn
from skimage.measure import regionprops
from sklearn.cluster import DBSCAN

HERE = os.path.dirname(os.path.abspath(__file__))


def centers_xy(G, R):
    """liso/utils/bev_utils.py:5-40 (get_metric_voxel_center_coords): pillar centres of a G x G grid over [-R/2, R/2)^2, float32"""
    ext = 0.5 * np.array([-R, -R, R, R], dtype=np.float64)
    size = ext[2:] - ext[:2]
    res = size / np.array([G, G])
    ii = np.arange(G)
    xs = ext[0] + (ii + 0.5) * res[0]
    ys = ext[1] + (ii + 0.5) * res[1]
    return np.stack(np.meshgrid(xs, ys, indexing="ij"), axis=-1).astype(np.float32)


def scene(seed, G, n_blobs, noise_cells):
    g = np.random.default_rng(seed)
    mask = np.zeros((G, G), bool)
    flow = np.zeros((G, G, 3), np.float32)
    rr, cc = np.meshgrid(np.arange(G), np.arange(G), indexing="ij")
    for _ in range(n_blobs):
        r0, c0 = g.uniform(0, G, 2)
        a, b, th = g.uniform(2.0, 14.0), g.uniform(1.5, 6.0), g.uniform(0, np.pi)
        u = (rr - r0) * np.cos(th) + (cc - c0) * np.sin(th)
        v = -(rr - r0) * np.sin(th) + (cc - c0) * np.cos(th)
        m = ((u / a) ** 2 + (v / b) ** 2 <= 1.0) & (g.random((G, G)) < g.uniform(0.35, 1.0))
        mask |= m
        flow[m] = g.normal(0, 0.6, 3).astype(np.float32) + g.normal(0, 0.05, (int(m.sum()), 3)).astype(np.float32)
    nz = g.integers(0, G, (noise_cells, 2))
    mask[nz[:, 0], nz[:, 1]] = True
    flow[nz[:, 0], nz[:, 1]] = g.normal(0, 1.0, (noise_cells, 3)).astype(np.float32)
    return mask, flow


def main():
    out = {"sklearn_version": np.array(sklearn.__version__), "skimage_version": np.array(skimage.__version__)}
    for tag, (G, nb, nz, seed) in {"a": (128, 12, 60, 0), "b": (256, 40, 200, 1), "c": (64, 3, 5, 2), "d": (96, 0, 30, 3), "e": (512, 60, 400, 4)}.items():
        R = G * 100.0 / 512.0
        ctr = centers_xy(G, R)
        mask, flow = scene(seed, G, nb, nz)
        lab_img = np.zeros(mask.shape, dtype=np.int64)
        if np.count_nonzero(mask) > 1:
            coords = np.concatenate([ctr[mask], 2.0 * flow[mask]], axis=-1)
            db = DBSCAN(eps=1.0, min_samples=5, metric="euclidean", algorithm="auto", n_jobs=1).fit(coords)
            lab = np.where(db.labels_ >= 0, db.labels_ + 1, 0)
            r_, c_ = np.nonzero(mask)
            lab_img[r_, c_] = lab
        props = np.array([[p.centroid[0], p.centroid[1], p.orientation, p.major_axis_length, p.minor_axis_length] for p in regionprops(lab_img.astype(np.int32))],
                         dtype=np.float64).reshape(-1, 5)
        out.update({f"{tag}_mask": mask, f"{tag}_flow": flow, f"{tag}_centers": ctr, f"{tag}_range": np.array(R), f"{tag}_labels": lab_img,
                    f"{tag}_props": props})
    dst = os.path.join(HERE, "dbscan_reference.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, sklearn.__version__, skimage.__version__, {k: int(out[k + "_labels"].max()) for k in "abcde"})


if __name__ == "__main__":
    main()
