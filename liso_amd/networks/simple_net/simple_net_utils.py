"""mirror of liso/networks/simple_net/simple_net_utils.py:8-40 (activations table, per-attribute channel counts)"""
from collections import OrderedDict

import torch

allowed_activations = {
    "none": lambda x: x,
    "softplus": torch.nn.functional.softplus,
    "sigmoid": torch.sigmoid,
    "tanh": torch.tanh,
    "exp": torch.exp,
}


def get_num_dims_per_box_attr(cfg):
    num_rot = {"direct": 1, "vector": 2, "none": 0, "class_bins": 36}[cfg.box_prediction.rotation_representation.method]
    num_dim = {"predict_aspect_ratio": 2, "predict_abs_size": 3, "predict_log_size": 3}[
        cfg.box_prediction.dimensions_representation.method]
    return OrderedDict(zip(("pos", "dims", "rot", "probs"),
                           (cfg.box_prediction.position_representation.num_box_pos_dims, num_dim, num_rot, 1)))
