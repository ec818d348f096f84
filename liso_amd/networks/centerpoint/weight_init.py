"""mirror of liso/networks/centerpoint/weight_init.py (initialisers used by RPN / SepHead)"""
import torch.nn as nn


def constant_init(module, val, bias=0):
    nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    assert distribution in ["uniform", "normal"]
    (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
    assert distribution in ["uniform", "normal"]
    fn = nn.init.kaiming_uniform_ if distribution == "uniform" else nn.init.kaiming_normal_
    fn(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)
