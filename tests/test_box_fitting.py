"""liso_amd/box_fitting/box_fitting.py against tests/golden/box_fitting_reference.npz, written by the reference's own four rectangle fits
(liso/box_fitting/box_fitting.py:10-258; tests/golden/make_box_fitting_golden.py): corners, angle, area and fit_2d_box_modest's
(centre, length, width, yaw) on 24 clusters.  Host functions: runs without a GPU."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "box_fitting_reference.npz"))


@pytest.mark.parametrize("method", ["min_zx_area_fit", "PCA", "variance_to_edge", "closeness_to_edge"])
def test_rectangle_fits_reproduce_the_reference(method):
    from liso_amd.box_fitting import box_fitting as bf

    fn = {"min_zx_area_fit": bf.minimum_bounding_rectangle, "PCA": bf.PCA_rectangle, "variance_to_edge": bf.variance_rectangle,
          "closeness_to_edge": bf.closeness_rectangle}[method]
    for i in range(int(G["n"])):
        pts, want = G[f"pts_{i}"], G[f"{method}_{i}"]
        corners, ang, area = fn(pts[:, [0, 1]])
        ctr, ln, wd, ry = bf.fit_2d_box_modest(pts, method)
        got = np.concatenate([np.asarray(corners).reshape(-1), [ang, area], ctr, [ln, wd, ry]])
        assert np.allclose(got, want, rtol=1e-9, atol=1e-9), (method, i, np.abs(got - want).max())


def test_unknown_method_raises():
    from liso_amd.box_fitting.box_fitting import fit_2d_box_modest

    with pytest.raises(NotImplementedError):
        fit_2d_box_modest(np.zeros((4, 3)), "best")
