"""Used ONLY by the tests/golden/make_*.py generators, in the build container where /root/reference exists.

Imports individual python files of the reference by path.  Third-party packages that are absent from this image
(mmcv, numba, munch, shapely, ...) are replaced by *registry / decorator / container* stubs that contain no
arithmetic: every number in a fixture is computed by the reference's own code and by torch/numpy.
"""
import importlib.util
import os
import sys
import types

REF = "/root/reference"


def _mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave like a package
        sys.modules[name] = m
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _identity_decorator_factory(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def install_stubs():
    import torch.nn as nn

    def build_norm_layer(cfg, num_features, postfix=""):
        cfg = dict(cfg)
        t = cfg.pop("type")
        layer = {"BN1d": nn.BatchNorm1d, "BN2d": nn.BatchNorm2d, "BN": nn.BatchNorm2d}[t](num_features, **cfg)
        return t, layer

    _mod("numba", jit=_identity_decorator_factory)
    _mod("mmcv")
    _mod("mmcv.cnn", build_norm_layer=build_norm_layer)
    _mod("mmcv.ops", DynamicScatter=object, Voxelization=object)
    _mod("mmcv.runner", force_fp32=_identity_decorator_factory, auto_fp16=_identity_decorator_factory)
    _mod("mmdet3d")
    _mod("mmdet3d.models")
    _mod("mmdet3d.models.builder", VOXEL_ENCODERS=_Registry(), MIDDLE_ENCODERS=_Registry())
    _mod("mmdet3d.models.voxel_encoders")
    _mod("mmdet3d.models.middle_encoders")
    _mod("mmdet3d.core")
    _mod("mmdet3d.core.voxel")


def load(name, relpath):
    """import /root/reference/<relpath> as module `name`"""
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    sys.dont_write_bytecode = True
    spec.loader.exec_module(m)
    return m


def load_mmdet3d_pillar_modules():
    install_stubs()
    base = "mmdetection3d/mmdet3d/"
    utils = load("mmdet3d.models.voxel_encoders.utils", base + "models/voxel_encoders/utils.py")
    enc = load("mmdet3d.models.voxel_encoders.pillar_encoder", base + "models/voxel_encoders/pillar_encoder.py")
    sc = load("mmdet3d.models.middle_encoders.pillar_scatter", base + "models/middle_encoders/pillar_scatter.py")
    vg = load("mmdet3d.core.voxel.voxel_generator", base + "core/voxel/voxel_generator.py")
    return utils, enc, sc, vg
