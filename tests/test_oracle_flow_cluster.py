"""CPU: pins oracle/flow_cluster.py against fixtures produced by the reference's own python."""
import os

import numpy as np
import torch

from oracle import flow_cluster as OF


def test_bev_dynamic_flow_and_zfit(golden_dir):
    g = np.load(os.path.join(golden_dir, "flow_cluster_reference.npz"))
    t = lambda k: torch.from_numpy(g[k])
    dyn, nrf = OF.bev_dynamic_flow(t("d1_valid"), t("d1_pcl"), t("d1_coors"), t("d1_flow"), t("d1_odom"), (64, 64))
    assert np.allclose(dyn.numpy(), g["d1_dyn"], rtol=1e-5, atol=1e-6)
    assert np.allclose(nrf.numpy(), g["d1_nrf"], rtol=1e-5, atol=1e-6)
    num, z, h = OF.fit_box_z(t("d3_pts"), t("d3_pos"), t("d3_dims"), t("d3_rot")[:, 0])
    assert np.array_equal(num.numpy(), g["d3_num"])
    assert np.allclose(z.numpy(), g["d3_z"], atol=1e-6) and np.allclose(h.numpy(), g["d3_h"], atol=1e-6)


def test_regionprops_restatement_on_analytic_shapes():
    """oracle.regionprops_restated against closed forms and the values scikit-image's own test-suite asserts for
    diagonal regions (skimage/measure/tests/test_regionprops.py::test_orientation: eye -> -pi/4, flipud(eye) -> +pi/4)"""
    import math

    import numpy as np

    from oracle.flow_cluster import regionprops_restated

    img = np.zeros((64, 64), int)
    img[10:30, 20:24] = 1   # 20 x 4 rectangle along the rows
    img[40:43, 5:55] = 2    # 3 x 50 rectangle along the columns
    p = regionprops_restated(img)
    assert np.allclose(p[0, :2], [19.5, 21.5]) and np.allclose(p[1, :2], [41.0, 29.5])
    assert abs(p[0, 2]) < 1e-12 and abs(abs(p[1, 2]) - math.pi / 2) < 1e-12
    for row, (L, W) in zip(p, [(20, 4), (50, 3)]):  # variance of L consecutive integers = (L^2 - 1) / 12
        assert np.allclose(row[3:], [4 * math.sqrt((L * L - 1) / 12), 4 * math.sqrt((W * W - 1) / 12)])
    d = np.eye(10, dtype=int)
    assert np.isclose(regionprops_restated(d)[0, 2], -math.pi / 4)
    assert np.isclose(regionprops_restated(np.flipud(d))[0, 2], math.pi / 4)


def test_dbscan_label_numbering_and_border_rule():
    """the two properties the device labelling relies on, checked on sklearn itself: clusters are numbered by their
    first core member in row-major order, and a border pillar between two clusters takes the smaller number"""
    import numpy as np

    from oracle.flow_cluster import dbscan_bev_labels

    G = 32
    xs = ((np.arange(G) + 0.5) * (100.0 / 512.0)).astype(np.float32)
    grid = np.stack(np.meshgrid(xs, xs, indexing="ij"), -1)
    for first_block_on_the_right in (False, True):
        mask = np.zeros((G, G), bool)
        flow = np.zeros((G, G, 3), np.float32)
        left, right = (slice(4, 8), slice(8, 12)), (slice(4, 8), slice(15, 19))
        if first_block_on_the_right:
            right = (slice(3, 8), slice(15, 19))  # starts one row earlier -> met first in row-major order
        mask[left], mask[right] = True, True
        flow[right + (0,)] = 0.9                   # 2 * 0.9 = 1.8 apart in feature space: the blocks never touch
        mask[5, 13], flow[5, 13, 0] = True, 0.45   # 0.9 from both in flow, 2 cells (0.39 m) from ONE cell of each block
        mask[20:24, 2:6] = True                    # a third cluster further down
        lab = dbscan_bev_labels(mask, flow, grid)
        l_left, l_right = int(lab[5, 8]), int(lab[5, 16])
        assert {l_left, l_right} == {1, 2} and lab[20, 2] == 3
        assert (l_right == 1) == first_block_on_the_right      # numbered by first member in row-major order
        assert lab[5, 13] == 1                                 # border pillar of both clusters -> the smaller number
