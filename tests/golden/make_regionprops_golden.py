"""Generates tests/golden/regionprops_reference.npz with scikit-image itself: skimage.measure.regionprops, the call
liso/networks/flow_cluster_detector/flow_cluster_detector.py:175-189 makes, on label images shaped like DBSCAN output.

scikit-image is not importable by this repository's interpreter (python 3.10, no wheel, no index), but the build container carries
a second interpreter with it: /opt/conda/bin/python3.9 (scikit-image 0.18.3, numpy 1.26, scipy 1.7; the reference pins 0.19.2 --
centroid / orientation / axis lengths are computed by the same code in both: `_regionprops.py` `centroid`, `orientation`,
`major_axis_length` / `minor_axis_length`, renamed `axis_major_length` / `axis_minor_length` in 0.19 with the old names kept).
Run in the build container only:  /opt/conda/bin/python3.9 tests/golden/make_regionprops_golden.py
"""
import os

import numpy as np
import skimage
from skimage.measure import regionprops

HERE = os.path.dirname(os.path.abspath(__file__))


def blobs(g, H, W, n, lo=3, hi=40):
    lab = np.zeros((H, W), np.int32)
    k = 0
    for _ in range(n):
        r, c = int(g.integers(0, H)), int(g.integers(0, W))
        cells = {(r, c)}
        for _ in range(int(g.integers(lo, hi))):  # random growth: ragged, sometimes with holes, sometimes thin
            rr, cc = list(cells)[int(g.integers(0, len(cells)))]
            dr, dc = [(0, 1), (1, 0), (0, -1), (-1, 0), (1, 1), (-1, 1)][int(g.integers(0, 6))]
            cells.add((min(max(rr + dr, 0), H - 1), min(max(cc + dc, 0), W - 1)))
        if any(lab[p] for p in cells):
            continue
        k += 1
        for p in cells:
            lab[p] = k
    return lab


def special():
    lab = np.zeros((40, 48), np.int32)
    lab[2, 3] = 1                       # a single pixel
    lab[5, 5:17] = 2                    # horizontal line
    lab[8:20, 30] = 3                   # vertical line
    for i in range(9):
        lab[22 + i, 4 + i] = 4          # main diagonal  (a == c: the orientation special case)
        lab[22 + i, 30 - i] = 5         # anti-diagonal  (a == c, b of the other sign)
    lab[10:14, 10:14] = 6               # square (a == c, b == 0)
    lab[30:33, 36:46] = 7               # 3 x 10 rectangle
    lab[34:39, 2:4] = 8                 # 5 x 2 rectangle
    lab[36, 20:23] = 9; lab[35:38, 21] = 9  # plus sign
    return lab


def main():
    g = np.random.default_rng(0)
    imgs = {"special": special(), "blobs_small": blobs(g, 64, 64, 25), "blobs_512": blobs(g, 512, 512, 120, 5, 400),
            "blobs_dense": blobs(g, 96, 128, 200, 2, 12)}
    out = {"skimage_version": np.array(skimage.__version__)}
    for name, lab in imgs.items():
        props = regionprops(lab)
        rows = np.array([[p.label, p.centroid[0], p.centroid[1], p.orientation, p.major_axis_length, p.minor_axis_length] for p in props],
                        dtype=np.float64)
        assert [int(r[0]) for r in rows] == list(range(1, int(lab.max()) + 1))  # regionprops lists the labels in ascending order
        out[name + "_labels"] = lab
        out[name + "_props"] = rows[:, 1:]
    dst = os.path.join(HERE, "regionprops_reference.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, skimage.__version__, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
