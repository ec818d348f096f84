"""Generates tests/golden/box_fitting_reference.npz from the reference's rectangle fits (liso/box_fitting/box_fitting.py:10-258: the four
criteria and fit_2d_box_modest) on synthetic clusters: L-shaped car returns at several headings, blobs, near-degenerate lines, few points.
Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_box_fitting_golden.py"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import import_with_stubs  # noqa: E402


def clusters():
    g = np.random.default_rng(5)
    out = []
    for i in range(24):
        yaw, l, w = g.uniform(-np.pi, np.pi), g.uniform(3.5, 5.0), g.uniform(1.6, 2.1)
        n = int(g.integers(8, 300))
        kind = i % 4
        if kind == 0:    # two visible sides of a car
            side = g.uniform(size=n) < 0.6
            p = np.where(side[:, None], np.stack([g.uniform(-l / 2, l / 2, n), np.full(n, -w / 2)], -1),
                         np.stack([np.full(n, l / 2), g.uniform(-w / 2, w / 2, n)], -1))
        elif kind == 1:  # filled rectangle
            p = np.stack([g.uniform(-l / 2, l / 2, n), g.uniform(-w / 2, w / 2, n)], -1)
        elif kind == 2:  # blob
            p = g.normal(0, [1.2, 0.5], (n, 2))
        else:            # thin line
            p = np.stack([g.uniform(-l / 2, l / 2, n), g.normal(0, 0.02, n)], -1)
        p = p + g.normal(0, 0.03, p.shape)
        R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        xy = p @ R.T + g.uniform(-30, 30, 2)
        out.append(np.concatenate([xy, g.uniform(-1.5, 0.3, (n, 1))], -1))
    return out


def main():
    def _imp():
        import liso.box_fitting.box_fitting as bf
        return bf

    bf = import_with_stubs(_imp)
    out = {}
    cl = clusters()
    out["n"] = np.array(len(cl))
    for i, c in enumerate(cl):
        out[f"pts_{i}"] = c
        for m, fn in (("min_zx_area_fit", bf.minimum_bounding_rectangle), ("PCA", bf.PCA_rectangle), ("variance_to_edge", bf.variance_rectangle),
                      ("closeness_to_edge", bf.closeness_rectangle)):
            corners, ang, area = fn(c[:, [0, 1]])
            ctr, ln, wd, ry = bf.fit_2d_box_modest(c, m)
            out[f"{m}_{i}"] = np.concatenate([np.asarray(corners).reshape(-1), [ang, area], ctr, [ln, wd, ry]])
    np.savez_compressed(os.path.join(HERE, "box_fitting_reference.npz"), **out)
    print("clusters", len(cl))


if __name__ == "__main__":
    sys.path.insert(0, "/root/reference")
    main()
