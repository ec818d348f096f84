"""Device-side mirrors of helpers in liso/datasets/torch_dataset_commons.py that sit on the mining / training path."""
from collections import abc, defaultdict

import numpy as np
import torch

from liso_amd.datasets.nuscenes.analyse_boxes import voxelize_pcl
from liso_amd.datasets.targets import render_center_targets  # noqa: F401  (draw_heat_regression_maps, :190-339)
from liso_amd.kabsch.shape_utils import Shape
from liso_amd.tracker.box_points import FP64_PRODUCT, dense_boxes, points_in_boxes


@torch.no_grad()
def get_points_in_boxes_mask(objects: Shape, pcl_homog, return_pcl_in_box_cosy=False, use_double_precision=True):
    """reference :1902-1935 -- bool [N,K]: point n lies inside box k.  `pcl_homog` [N,4] homogeneous (last column 1), boxes
    unbatched [K]; fp64 transform as in the reference's torch branch."""
    assert pcl_homog.shape[-1] == 4 and len(pcl_homog.shape) == 2, pcl_homog.shape
    assert torch.is_tensor(pcl_homog), "device path only (the numpy branch of the reference is data-loader code)"
    assert use_double_precision, "not implemented for torch"  # reference :1912
    res = points_in_boxes(dense_boxes(objects)[None], pcl_homog[None, :, :3], want_mask=True, want_count=False,
                          precision=FP64_PRODUCT)
    if return_pcl_in_box_cosy:
        # reference :1914-1918 -- the points in every box's frame, [N,K,4] in the cloud's dtype (fp64 product, rounded once): the one
        # caller-visible tensor of size N x K; callers that only need the mask never pay for it
        box_T_sensor = torch.linalg.inv_ex(objects.get_poses()).inverse
        pcl_box = torch.einsum("kij,nj->nki", box_T_sensor, pcl_homog.to(box_T_sensor.dtype)).to(pcl_homog.dtype)
        return res["mask"][0], pcl_box
    return res["mask"][0]


def voxelize_sample(pcl, bev_range_m, img_grid_size, height_range_m=(-np.inf, np.inf)):
    """`LidarDataset.voxelize_sample` (reference :975-987) as a function: points [N,3+] (numpy, or a torch tensor on any
    device) -> (pillar_coors int32 [N,2], point_is_in_range bool [N]).  Same dtype path as the reference: the float32 BEV
    range is extended by a float64 z range of 1000 m and the int32 grid by an int64 1, so float32 points are promoted to
    float64 before the division, the scaling and the truncating int32 conversion."""
    rng3 = np.append(np.asarray(bev_range_m, np.float32), np.array(1000.0))
    grid3 = np.append(np.asarray(img_grid_size).astype(np.int32), np.array(1))
    hr = np.asarray(height_range_m, np.float32)
    if torch.is_tensor(pcl):
        coors, inside = voxelize_pcl(pcl, torch.from_numpy(rng3).to(pcl.device), torch.from_numpy(grid3).to(pcl.device))
        in_h = (float(hr[0]) < pcl[:, 2]) & (pcl[:, 2] < float(hr[1]))
    else:
        coors, inside = voxelize_pcl(pcl, rng3, grid3)
        in_h = (hr[0] < pcl[:, 2]) & (pcl[:, 2] < hr[1])
    return coors[..., 0:2], inside & in_h


# ---- collate (reference :340-431): list of per-sample dicts -> batched dict with NaN / -1 padding ------------------------
def _list_of_dict_to_dict_of_list(in_list):
    if all(torch.is_tensor(el) for el in in_list):
        return in_list
    res = defaultdict(list)
    for sub in in_list:
        for key in sub:
            res[key].append(sub[key])
    return {k: (dict(v) if isinstance(v, defaultdict) else v) for k, v in res.items()}


def change_k_v(key, parent_dict, value):
    """reference :380-431"""
    if key in ("pcl_ta", "pcl_tb", "pcl_tx") and isinstance(value, abc.Mapping):
        pcls = torch.nn.utils.rnn.pad_sequence(value["pcl"], batch_first=True, padding_value=np.nan)
        coors = torch.nn.utils.rnn.pad_sequence(value["pillar_coors"], batch_first=True, padding_value=-1)
        mask = torch.logical_not(torch.isnan(pcls).sum(-1))
        parent_dict[key] = {"pcl": pcls, "pcl_is_valid": mask, "pillar_coors": coors}
        return
    if isinstance(value, abc.Mapping):
        for sub_key, sub_value in list(value.items()):
            change_k_v(sub_key, value, sub_value)
        return
    if key in ("pillar_coors", "pcl"):
        return  # processed jointly above
    if key in ("moving_mask", "point_has_valid_flow_label"):
        parent_dict[key] = torch.nn.utils.rnn.pad_sequence(parent_dict[key], batch_first=True, padding_value=False)
    elif all(isinstance(v, Shape) for v in value):
        parent_dict[key] = Shape.from_list_of_shapes(value)
    elif key in ("flow_ta_tb", "flow_tb_ta"):
        parent_dict[key] = torch.nn.utils.rnn.pad_sequence(parent_dict[key], batch_first=True, padding_value=np.nan)
    elif "pcl_full" in key or "lidar_rows" in key:
        parent_dict[key] = value
    elif "track_ids_mask" in key:
        parent_dict[key] = torch.nn.utils.rnn.pad_sequence(parent_dict[key], batch_first=True, padding_value=0)
    else:
        parent_dict[key] = torch.stack(value, dim=0)


def collate_list_data(samples):
    """reference :370-377 -- the batch layout every hot-path op consumes: `pcl_ta` = {pcl [B,Nmax,C] NaN-padded,
    pcl_is_valid [B,Nmax], pillar_coors [B,Nmax,2] padded with -1}, flows NaN-padded, `pcl_full_*` kept as lists."""
    out = _list_of_dict_to_dict_of_list(samples)
    if any(isinstance(i, abc.Mapping) for v in out.values() for i in v):
        out = {k: _list_of_dict_to_dict_of_list(v) for k, v in out.items()}
    for k, v in list(out.items()):
        change_k_v(k, out, v)
    return out
