"""Generates tests/golden/flow_io_reference.npz from the reference's own python (the SLIM flow ingest):
  liso.datasets.torch_dataset_commons.LidarDataset.expand_valid_bev_flow_to_zero_flow_neighbor_pillars   (:677-695)
  liso.datasets.torch_dataset_commons.LidarDataset.load_add_flow_to_sample_content                       (:590-675)
Both are methods; they are called unbound on a namespace object that carries exactly the attributes they read (prediction path,
a loader that is numpy's np.load, cfg.data.flow_source / augmentation, bev_range_m_np).  The flow file they read is written
here in the format slim/experiment.py:389-404,460-468 exports.  Third-party imports of the module that are absent from this
image are stubbed with empty modules (names only).  Run in the build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_flow_io_golden.py
"""
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_targets_golden import _Anything, cfg, import_with_stubs  # noqa: E402  (also puts /root/reference on sys.path)

sys.modules["torch.utils.tensorboard"] = _Anything("torch.utils.tensorboard")


def flow_map(g, G, fill):
    f = np.zeros((G, G, 2), np.float32)
    m = g.random((G, G)) < fill
    f[m] = g.normal(0, 1, (int(m.sum()), 2)).astype(np.float32)
    f[0, :5] = 1.5    # np.roll wraps around at the border
    f[:, -1] = 0.0
    f[-1, 3:9] = -0.7
    return f


def main():
    def _imp():
        from liso.datasets.torch_dataset_commons import LidarDataset
        return LidarDataset

    LidarDataset = import_with_stubs(_imp)
    g = np.random.default_rng(0)
    out = {}
    for tag, (G, fill) in {"a": (32, 0.35), "b": (64, 0.1), "c": (48, 0.8)}.items():
        f = flow_map(g, G, fill)
        out[f"expand_{tag}_in"] = f
        out[f"expand_{tag}_out"] = np.asarray(LidarDataset.expand_valid_bev_flow_to_zero_flow_neighbor_pillars(None, f.copy()))
    # the whole ingest for one sample pair, both directions
    G, R = 64, 80.0
    content = {"bev_raw_flow_t0_t1": flow_map(g, G, 0.4), "bev_raw_flow_t1_t0": flow_map(g, G, 0.4), "bev_range_m": np.array([R, R])}
    pcl = {k: np.concatenate([g.uniform(-0.6 * R, 0.6 * R, (4000, 2)), g.uniform(-2, 2, (4000, 1)), g.random((4000, 1))], -1).astype(np.float32)
           for k in ("t0", "t1")}  # some points lie outside the flow grid: they take the mean flow of the inside points
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / "sample_000.npz"
        np.savez(path, **content)
        fake = types.SimpleNamespace(
            pred_flow_path=Path(tmp), loader_saver_helper=types.SimpleNamespace(load_sample=lambda p, fn, **kw: dict(fn(p, **kw))),
            cfg=cfg({"data": {"flow_source": "slim_bev_120m", "augmentation": {"active": False}}}), use_geom_augmentation=False,
            bev_range_m_np=np.array([60.0, 60.0]),
            expand_valid_bev_flow_to_zero_flow_neighbor_pillars=lambda bf: LidarDataset.expand_valid_bev_flow_to_zero_flow_neighbor_pillars(None, bf))
        sample = {"pcl_t0": pcl["t0"].copy(), "pcl_t1": pcl["t1"].copy()}
        LidarDataset.load_add_flow_to_sample_content(fake, "sample_000.bin", sample, "t0", "t1")  # (the reference calls it with these keys: kitti_raw_torch_dataset.py:246-248)
    out.update({"ingest_bev_t0_t1": content["bev_raw_flow_t0_t1"], "ingest_bev_t1_t0": content["bev_raw_flow_t1_t0"],
                "ingest_bev_range_m": content["bev_range_m"], "ingest_pcl_t0": pcl["t0"], "ingest_pcl_t1": pcl["t1"],
                "ingest_flow_t0_t1": np.asarray(sample["slim_bev_120m"]["flow_t0_t1"]),
                "ingest_flow_t1_t0": np.asarray(sample["slim_bev_120m"]["flow_t1_t0"])})
    dst = os.path.join(HERE, "flow_io_reference.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
