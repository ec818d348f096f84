"""SLIM flow ingest on the device: the same tensor ops as tests/test_flow_io.py, inputs resident in HBM,
against the numpy masked-array restatement; plus the export dictionary built from device predictions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_point_lookup_and_neighbour_filling_on_the_device():
    from liso_amd.slim import flow_io
    from oracle.flow_io import expand_valid_bev_flow_to_zero_flow_neighbor_pillars as ref_expand, point_flow_from_bev as ref_lookup

    dev = torch.device("cuda")
    for seed, G, n in ((0, 64, 5000), (1, 512, 120000)):
        g = np.random.default_rng(seed)
        f = np.zeros((G, G, 2), np.float32)
        m = g.random((G, G)) < 0.35
        f[m] = g.normal(0, 1, (int(m.sum()), 2)).astype(np.float32)
        f[0, :5] = 1.5
        got_e = flow_io.expand_valid_bev_flow_to_zero_flow_neighbor_pillars(torch.from_numpy(f).to(dev))
        assert got_e.is_cuda and np.array_equal(got_e.cpu().numpy(), ref_expand(f))
        pcl = np.concatenate([g.uniform(-60, 60, (n, 2)), g.uniform(-2, 2, (n, 1)), g.random((n, 1))], -1).astype(np.float32)
        rng = np.array([100.0, 100.0])
        got = flow_io.point_flow_from_bev(torch.from_numpy(pcl).to(dev), torch.from_numpy(f).to(dev), rng)
        want = ref_lookup(pcl, f, rng)
        assert got.is_cuda and got.shape == (n, 3)
        assert np.allclose(got.cpu().numpy(), want, rtol=1e-5, atol=1e-6)


def test_export_dictionary_from_device_predictions():
    from liso_amd.slim import flow_io
    from liso_amd.utils.config import AttrDict

    dev = torch.device("cuda")
    G = 64
    mk = lambda s: AttrDict(modified_network_output=AttrDict(static_flow=torch.full((1, G, G, 2), float(s), device=dev),  # noqa: E731
                                                              dynamicness=torch.full((1, G, G), 0.1 * s, device=dev)))
    content = flow_io.flow_export_dict([mk(1), mk(2)], [mk(3), mk(4)], torch.tensor(0.37, device=dev), np.array([100.0, 100.0]))
    assert all(isinstance(v, np.ndarray) for v in content.values())  # the wire format is numpy (experiment.py:389-404)
    assert float(content["bev_raw_flow_t0_t1"][0, 0, 0]) == 2.0 and float(content["bev_raw_flow_t1_t0"][3, 3, 1]) == 4.0
    assert abs(float(content["static_threshold"]) - 0.37) < 1e-6


def test_device_ingest_matches_the_reference_fixture():
    """the fixture written by the reference's LidarDataset methods (tests/golden/make_flow_io_golden.py), inputs resident in HBM"""
    import os

    from liso_amd.slim import flow_io

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "flow_io_reference.npz"))
    dev = torch.device("cuda")
    for tag in "abc":
        got = flow_io.expand_valid_bev_flow_to_zero_flow_neighbor_pillars(torch.from_numpy(fx[f"expand_{tag}_in"]).to(dev))
        assert np.array_equal(got.cpu().numpy(), fx[f"expand_{tag}_out"])
    for a, b in (("t0", "t1"), ("t1", "t0")):
        got = flow_io.point_flow_from_bev(torch.from_numpy(fx[f"ingest_pcl_{a}"]).to(dev), torch.from_numpy(fx[f"ingest_bev_{a}_{b}"]).to(dev),
                                          fx["ingest_bev_range_m"])
        assert np.allclose(got.cpu().numpy(), fx[f"ingest_flow_{a}_{b}"], rtol=1e-5, atol=1e-6)


def test_two_direction_export_inference_equals_the_full_forward():
    """SLIM.infer_export_predictions (both flow directions as one batch, last RAFT iteration, one dense decode per direction) gives the
    arrays of the reference's export -- bev_raw_flow_t0_t1 AND bev_raw_flow_t1_t0, dynamicness both ways (experiment.py:389-404) --
    exactly as the evaluation forward's last iteration does"""
    from liso_amd.datasets.synthetic import slim_pair
    from liso_amd.slim.flow_io import flow_export_dict
    from liso_amd.slim.model.slim import SLIM
    from liso_amd.utils.config import default_cfg

    dev = torch.device("cuda")
    torch.manual_seed(4)
    cfg = default_cfg(grid=256, bev_range_m=50.0)
    net = SLIM(cfg, 100).to(dev).eval()
    s0, s1 = slim_pair(21, dev, n_points=30000, grid=256, bev_range_m=50.0)
    with torch.no_grad():
        fw, bw = net(s0, s1, None)
        efw, ebw = net.infer_export_predictions(s0, s1)
    thr = net.moving_dynamicness_threshold.value()
    full = flow_export_dict(fw, bw, thr, cfg.data.bev_range_m)
    fast = flow_export_dict([efw], [ebw], thr, cfg.data.bev_range_m)
    assert set(full) == set(fast) and {"bev_raw_flow_t0_t1", "bev_raw_flow_t1_t0", "bev_dynamicness_t0_t1", "bev_dynamicness_t1_t0"} <= set(fast)
    for k in full:
        a, b = np.asarray(full[k], np.float64), np.asarray(fast[k], np.float64)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-5 * max(np.abs(a).max(), 1e-6), k
