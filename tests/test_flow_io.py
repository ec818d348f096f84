"""SLIM flow wire format: ingest arithmetic vs the numpy masked-array restatement of the reference,
and the npz keys a reference reader expects.  Pure host code: runs on CPU."""
import numpy as np
import torch

from liso_amd.slim import flow_io
from liso_amd.utils.config import AttrDict


def _flow_map(seed, G=64):
    g = np.random.default_rng(seed)
    f = np.zeros((G, G, 2), np.float32)
    m = g.random((G, G)) < 0.35
    f[m] = g.normal(0, 1, (int(m.sum()), 2)).astype(np.float32)
    f[0, :5] = 1.5      # exercises np.roll's wrap-around at the border
    f[:, -1] = 0.0
    return f


def test_expand_and_point_lookup_match_the_masked_array_restatement():
    from oracle.flow_io import expand_valid_bev_flow_to_zero_flow_neighbor_pillars as ref_expand, point_flow_from_bev as ref_lookup

    for seed in range(3):
        f = _flow_map(seed)
        assert np.array_equal(flow_io.expand_valid_bev_flow_to_zero_flow_neighbor_pillars(f), ref_expand(f))
        g = np.random.default_rng(100 + seed)
        pcl = np.concatenate([g.uniform(-30, 30, (5000, 2)), g.uniform(-2, 2, (5000, 1)), g.random((5000, 1))], -1).astype(np.float32)
        rng = np.array([50.0, 50.0])  # some points fall outside the flow grid -> mean flow of the inside points
        got = flow_io.point_flow_from_bev(pcl, f, rng)
        want = ref_lookup(pcl, f, rng)
        assert got.shape == (5000, 3) and np.allclose(got, want, rtol=1e-6, atol=1e-7)
        got_t = flow_io.point_flow_from_bev(torch.from_numpy(pcl), torch.from_numpy(f), rng)
        assert torch.is_tensor(got_t) and np.allclose(got_t.numpy(), want, rtol=1e-6, atol=1e-7)


def test_export_keys_and_round_trip(tmp_path):
    G = 32
    mk = lambda s: AttrDict(modified_network_output=AttrDict(static_flow=torch.full((1, G, G, 2), float(s)),  # noqa: E731
                                                              dynamicness=torch.full((1, G, G), 0.1 * s)))
    preds_fw, preds_bw = [mk(1), mk(2)], [mk(3), mk(4)]
    content = flow_io.flow_export_dict(preds_fw, preds_bw, torch.tensor(0.37), np.array([100.0, 100.0]))
    assert set(content) == {"bev_raw_flow_t0_t1", "bev_raw_flow_t1_t0", "bev_dynamicness_t0_t1", "bev_dynamicness_t1_t0",
                            "static_threshold", "bev_range_m"}  # experiment.py:389-404,460-468
    assert content["bev_raw_flow_t0_t1"].shape == (G, G, 2) and float(content["bev_raw_flow_t0_t1"][0, 0, 0]) == 2.0  # LAST iteration
    assert float(content["bev_raw_flow_t1_t0"][0, 0, 0]) == 4.0 and content["bev_dynamicness_t0_t1"].shape == (G, G)
    path = tmp_path / "seq" / "sample_000.npz"
    flow_io.save_flow_npz(path, content)
    back = flow_io.load_flow_npz(path)
    assert all(np.array_equal(back[k], content[k]) for k in content)
    sample = {"pcl_ta": np.random.default_rng(0).uniform(-40, 40, (100, 4)).astype(np.float32),
              "pcl_tb": np.random.default_rng(1).uniform(-40, 40, (100, 4)).astype(np.float32)}
    flow_io.add_flow_to_sample(sample, back, "slim_bev_120m")
    assert sample["slim_bev_120m"]["flow_ta_tb"].shape == (100, 3) and np.allclose(sample["slim_bev_120m"]["flow_ta_tb"][:, :2], 2.0)
    assert np.allclose(sample["slim_bev_120m"]["flow_tb_ta"][:, :2], 4.0)


def _fixture():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "flow_io_reference.npz"))


def test_oracle_and_host_ops_match_the_reference_fixture():
    """tests/golden/flow_io_reference.npz was written by the reference's own LidarDataset methods
    (tests/golden/make_flow_io_golden.py): the numpy restatement (oracle/flow_io.py) and the tensor ops reproduce it."""
    from oracle.flow_io import expand_valid_bev_flow_to_zero_flow_neighbor_pillars as ref_expand, point_flow_from_bev as ref_lookup

    fx = _fixture()
    for tag in "abc":
        f, want = fx[f"expand_{tag}_in"], fx[f"expand_{tag}_out"]
        assert np.array_equal(ref_expand(f.copy()), want)
        assert np.array_equal(flow_io.expand_valid_bev_flow_to_zero_flow_neighbor_pillars(f.copy()), want)
    rng = fx["ingest_bev_range_m"]
    for a, b in (("t0", "t1"), ("t1", "t0")):
        pcl, bev, want = fx[f"ingest_pcl_{a}"], fx[f"ingest_bev_{a}_{b}"], fx[f"ingest_flow_{a}_{b}"]
        assert np.allclose(ref_lookup(pcl, bev, rng), want, rtol=1e-6, atol=1e-7)
        assert np.allclose(flow_io.point_flow_from_bev(pcl, bev, rng), want, rtol=1e-6, atol=1e-7)
    # the whole ingest through add_flow_to_sample, keys as the reference's datasets call it (src "t0", target "t1")
    sample = {"pcl_t0": fx["ingest_pcl_t0"], "pcl_t1": fx["ingest_pcl_t1"]}
    pred = {"bev_raw_flow_t0_t1": fx["ingest_bev_t0_t1"], "bev_raw_flow_t1_t0": fx["ingest_bev_t1_t0"], "bev_range_m": rng}
    flow_io.add_flow_to_sample(sample, pred, "slim_bev_120m", src_key="t0", target_key="t1")
    assert np.allclose(sample["slim_bev_120m"]["flow_t0_t1"], fx["ingest_flow_t0_t1"], rtol=1e-6, atol=1e-7)
    assert np.allclose(sample["slim_bev_120m"]["flow_t1_t0"], fx["ingest_flow_t1_t0"], rtol=1e-6, atol=1e-7)
