"""torch's convolution for HOST tensors -- the only call site of a library convolution in the package.

The modules of liso_amd run on the own gfx950 kernels (liso_amd/utils/mfma_conv.py -> include/liso_conv.h) whenever their input is on
the GPU; there is no switch that routes device tensors anywhere else, and a device tensor the kernels do not cover raises
(`mfma_conv.on_device`).  What remains is the host-logic test tier (`pytest -m "not gpu"`: state-dict layout, trainer plumbing,
gloo data parallelism at world sizes 2 and 4), which steps the same modules on CPU tensors: those convolutions come here.
Every function refuses device tensors.  (A library-backed comparison run -- scripts/compare_backends.py -- lifts that refusal for
its own process; nothing in the package or in bench.py does.)
"""
import torch
import torch.nn.functional as F

_HOST_ONLY = True  # scripts/compare_backends.py only


def _check(x):
    if _HOST_ONLY and x.is_cuda:
        raise RuntimeError("liso_amd.utils.host_ops: device tensors run on the library's own kernels (liso_amd/libliso_hip.so); "
                           "this is the host path of the CPU test tier")


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1):
    _check(x)
    return F.conv2d(x, w, b, stride, padding, dilation)


def conv_transpose2d(x, w, b=None, stride=1, padding=0):
    _check(x)
    return F.conv_transpose2d(x, w, b, stride=stride, padding=padding)


def module_forward(layer, x):
    """layer(x) for an nn.Conv2d / nn.ConvTranspose2d on a host tensor"""
    _check(x)
    return layer(x)
