"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute without a GPU)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        syms |= set(re.findall(r"\b(liso_[a-z0-9_]+)\s*\(", txt))
    return syms


def test_library_exports_every_declared_symbol():
    from liso_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = _lib.lib()
    declared = _declared_symbols()
    assert declared, "no declarations parsed"
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"
    # and the ctypes table covers exactly the declared ABI
    assert set(_lib.SIGNATURES) == declared


def test_host_entry_point_matches_oracle():
    # boxes_iou_bev_cpu is the reference's explicitly-CPU API (iou3d_cpu.cpp:232-252): callable without a GPU
    import torch

    from liso_amd import iou3d_nms_cuda as M
    from oracle import iou3d as O

    a, _ = O.random_boxes(120, 4, 8.0)
    b, _ = O.random_boxes(90, 5, 8.0)
    out = torch.zeros(120, 90)
    assert M.boxes_iou_bev_cpu(torch.from_numpy(a), torch.from_numpy(b), out) == 1
    assert np.array_equal(out.numpy().view(np.uint32), O.boxes_iou_bev(a, b).view(np.uint32))


def test_device_ops_refuse_cpu_tensors():
    import torch

    from liso_amd import _lib, iou3d_nms_cuda as M

    with pytest.raises(_lib.LisoHipError):
        M.boxes_iou_bev_gpu(torch.zeros(2, 7), torch.zeros(2, 7), torch.zeros(2, 2))
    with pytest.raises(_lib.LisoHipError):
        M.nms_gpu(torch.zeros(2, 7), torch.zeros(2, dtype=torch.int64), 0.1)


def test_error_codes_without_gpu():
    from liso_amd import _lib

    lib = _lib.lib()
    assert lib.liso_iou3d_nms_workspace_bytes(1000) == 1000 * 16 * 8
    assert lib.liso_iou3d_nms_workspace_bytes(0) == 0
    assert lib.liso_iou3d_iou_bev_f32(None, -1, None, 3, None, None) == -1
    assert lib.liso_iou3d_iou_bev_f32(None, 0, None, 3, None, None) == 0  # empty is fine, nothing launched
