"""CenterPointStyleNet.  Mirror of liso/networks/simple_net/centerpoint_net.py (same config keys, sub-module names
`pfn`, `rpn`, `center_head`; forward(img_t0, pcls) -> (dict of NHWC maps, aux))."""
import torch

from liso_amd.networks.centerpoint.center_head import CenterHead
from liso_amd.networks.centerpoint.rpn import RPN
from liso_amd.networks.pcl_to_feature_grid.pcl_to_feature_grid import PointsPillarFeatureNetWrapper
from liso_amd.networks.simple_net.simple_net_utils import get_num_dims_per_box_attr


def is_power_of_two(n):
    return (n != 0) and (n & (n - 1) == 0)


class CenterPointStyleNet(torch.nn.Module):
    def __init__(self, cfg) -> None:
        super().__init__()
        self.cfg = cfg
        crf = cfg.network.centerpoint.setdefault("channel_reduction_factor", 1)
        assert is_power_of_two(crf), crf
        cp = cfg.network.centerpoint
        rpn_conf = {  # reference :22-35
            "layer_nums": [3, 5], "ds_layer_strides": [2, 2],
            "ds_num_filters": [cp.hid_dim // crf, 128 // crf],
            "us_layer_strides": [0.5, 1], "us_num_filters": [128 // crf, 128 // crf],
        }
        if cp.reduce_receptive_field == 2:
            rpn_conf["ds_layer_strides"] = [1, 1]
        elif cp.reduce_receptive_field == 1:
            rpn_conf["ds_layer_strides"] = [1, 2]
        elif cp.reduce_receptive_field != 0:
            raise NotImplementedError(cp.reduce_receptive_field)
        if cp.use_baseline_parameters:  # reference :46-59
            rpn_conf["layer_nums"].append(5)
            rpn_conf["ds_layer_strides"].append(2)
            rpn_conf["ds_num_filters"].append(256 // crf)
            rpn_conf["us_layer_strides"].append(2)
            rpn_conf["us_num_filters"].append(128 // crf)
            head_conf = {"stride": 1, "in_channels": sum(rpn_conf["us_num_filters"])}
        else:
            head_conf = {"stride": 2, "in_channels": sum(rpn_conf["us_num_filters"])}
        self.pfn = PointsPillarFeatureNetWrapper(cfg)
        self.rpn = RPN(**rpn_conf, num_input_features=cp.hid_dim // crf, norm_cfg=dict(cp.batch_norm.kwargs))
        assert cfg.box_prediction.rotation_representation.method in ("vector", "class_bins")
        common_heads = {k: (v, 2) for k, v in get_num_dims_per_box_attr(cfg).items()}
        self.center_head = CenterHead(**head_conf, common_heads=common_heads, norm_cfg=dict(cp.batch_norm.kwargs))

    def set_compute_dtype(self, dtype):
        """fp32 = parity configuration; bf16 = BASELINE config 3 (bf16 BEV tensors, fp32 BN stats / head outputs)."""
        self.pfn.out_dtype = dtype

    def forward(self, img_t0, pcls, canvas=None):
        """`canvas` (extension): precomputed (bev_enc, occupancy) of `self.pfn` -- callers that replay backbone + head from a
        hipGraph keep the pillar encoder outside of it"""
        bev_enc, bev_occupancy_map = canvas if canvas is not None else self.pfn(pcl_t0=pcls, img_t0=img_t0)
        aux_outputs = {"bev_net_input_dbg": bev_occupancy_map}
        # (the occupancy map lets the first RPN layer multiply occupied pillar cells only: liso_amd/utils/mfma_conv.py `_sparse_stem`)
        pred_dict = self.center_head(self.rpn(bev_enc, lazy=True, occupancy=bev_occupancy_map))
        return {k: v.permute(0, 2, 3, 1) for k, v in pred_dict.items()}, aux_outputs  # reference :111
