"""mirror of liso/networks/centerpoint/norm.py:55-56 (the only builder the hot path uses)"""
from torch import nn


def baurst_build_norm_layer(norm_cfg, num_features):
    return None, nn.BatchNorm2d(num_features=num_features, **norm_cfg)
