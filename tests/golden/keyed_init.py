"""Deterministic network initialisation that depends only on the state_dict KEY NAMES and shapes (not on module construction
order or on a global RNG), so that a fixture generator driving the reference's modules and a test driving the product's
modules start from the same weights without storing ~19 MB of them.  Pure helper: no reference code involved."""
import zlib

import torch


def keyed_state_dict(shapes):
    """shapes: {key: (shape tuple, torch dtype)} -> {key: tensor}; conv / linear weights ~ N(0, 2/fan_in), BatchNorm weight
    ~ U(0.5, 1.5), BatchNorm bias and conv bias ~ U(-0.2, 0.2), running_mean 0 / running_var 1 / counters 0; every other
    tensor (non-trainable constants such as `pillar_center_coors_m`) is NOT produced -- keep the module's own value."""
    out = {}
    for k, (shape, dtype) in shapes.items():
        g = torch.Generator().manual_seed(zlib.crc32(k.encode()) & 0x7FFFFFFF)
        leaf = k.split(".")[-1]
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros(shape, dtype=dtype)
        elif leaf == "running_mean":
            out[k] = torch.zeros(shape, dtype=dtype)
        elif leaf == "running_var":
            out[k] = torch.ones(shape, dtype=dtype)
        elif leaf == "weight" and len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            out[k] = (torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5).to(dtype)
        elif leaf == "weight":
            out[k] = (torch.rand(shape, generator=g) + 0.5).to(dtype)
        elif leaf == "bias":
            out[k] = (torch.rand(shape, generator=g) * 0.4 - 0.2).to(dtype)
    return out


def sample_indices(key, numel, n=64):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ 0x5A5A5A) & 0x7FFFFFFF)
    return torch.randint(0, numel, (min(n, numel),), generator=g)
