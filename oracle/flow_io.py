"""TEST INFRASTRUCTURE ONLY: numpy restatement of the reference's flow ingest, written with the
same numpy masked-array operations as liso/datasets/torch_dataset_commons.py:627-688 so that the product's tensor
formulation is checked against the reference's own semantics.  No reference fixture exists for this block (it needs the
dataset loader): parity unpinned beyond this restatement."""
import numpy as np


def expand_valid_bev_flow_to_zero_flow_neighbor_pillars(bev_flow):
    """torch_dataset_commons.py:670-688"""
    zero = (bev_flow == 0.0).all(axis=-1)
    m = np.ma.masked_array(bev_flow.copy(), mask=np.stack([zero, zero], axis=-1))
    for shift in (-1, 1):
        for axis in (0, 1):
            sh = np.roll(m, shift=shift, axis=axis)
            idx = ~sh.mask * m.mask
            m[idx] = sh[idx]
    return m.filled(fill_value=0.0)


def point_flow_from_bev(pcl, bev_flow, bev_range_m):
    """torch_dataset_commons.py:616-667 (voxelize_pcl: analyse_boxes.py:6-26)"""
    rng = np.append(bev_range_m, np.array(1000.0))
    grid = np.append(bev_flow.shape[:2], np.array(1))
    coors = ((pcl[:, :3] + 0.5 * rng) / rng * grid).astype(np.int32)
    ok = ((0 <= coors) & (coors < grid)).all(axis=1)
    flow2d = np.nan * np.ones((pcl.shape[0], 2), dtype=np.float32)
    f = expand_valid_bev_flow_to_zero_flow_neighbor_pillars(bev_flow)
    flow2d[ok] = f[coors[ok][:, 0], coors[ok][:, 1]]
    flow2d[~ok] = np.mean(flow2d[ok], axis=0)
    return np.concatenate([flow2d, np.zeros_like(flow2d[:, :1])], axis=-1)
