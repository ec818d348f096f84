"""Device-side mirrors of helpers in liso/datasets/torch_dataset_commons.py that sit on the mining / training path."""
import torch

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
    if return_pcl_in_box_cosy:
        raise NotImplementedError("return_pcl_in_box_cosy materialises [N,K,4]; not provided by the fused kernel")
    res = points_in_boxes(dense_boxes(objects)[None], pcl_homog[None, :, :3], want_mask=True, want_count=False,
                          precision=FP64_PRODUCT)
    return res["mask"][0]
