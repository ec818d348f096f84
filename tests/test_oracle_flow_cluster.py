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
