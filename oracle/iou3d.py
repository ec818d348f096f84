"""TEST INFRASTRUCTURE ONLY: numpy front-end of oracle/iou3d_oracle.c and of oracle/_ref.

Functions mirror the reference module iou3d_nms_cuda (iou3d_nms/src/iou3d_nms_api.cpp:11-17).
"""
import ctypes

import numpy as np

from . import load_oracle, load_ref

_fp = ctypes.POINTER(ctypes.c_float)
_lp = ctypes.POINTER(ctypes.c_int64)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 7, a.shape
    return a


def _pair(fn, a, b):
    a, b = _f(a), _f(b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    if out.size:
        fn(a.ctypes.data_as(_fp), a.shape[0], b.ctypes.data_as(_fp), b.shape[0], out.ctypes.data_as(_fp))
    return out


def boxes_iou_bev(a, b):
    """iou3d_cpu.cpp:232-252"""
    return _pair(load_oracle().oracle_boxes_iou_bev, a, b)


def boxes_overlap_bev(a, b):
    """iou3d_nms_kernel.cu:236-249 with host arithmetic (iou3d_cpu.cpp:128-220)"""
    return _pair(load_oracle().oracle_boxes_overlap_bev, a, b)


def _nms(fn, boxes, thresh):
    boxes = _f(boxes)
    keep = np.zeros(boxes.shape[0], np.int64)
    fn.restype = ctypes.c_int
    n = fn(boxes.ctypes.data_as(_fp), boxes.shape[0], ctypes.c_float(thresh), keep.ctypes.data_as(_lp))
    return keep[:n].copy()


def nms(boxes_sorted, thresh):
    """iou3d_nms.cpp:90-136 (boxes already sorted by descending score) -> kept indices"""
    return _nms(load_oracle().oracle_nms, boxes_sorted, thresh)


def nms_normal(boxes_sorted, thresh):
    """iou3d_nms.cpp:139-186"""
    return _nms(load_oracle().oracle_nms_normal, boxes_sorted, thresh)


def nms_from_iou(iou, thresh):
    """greedy sweep of iou3d_nms.cpp:113-132 over a given IoU matrix"""
    iou = np.ascontiguousarray(iou, np.float32)
    n = iou.shape[0]
    keep = np.zeros(n, np.int64)
    fn = load_oracle().oracle_nms_from_iou
    fn.restype = ctypes.c_int
    k = fn(iou.ctypes.data_as(_fp), n, ctypes.c_float(thresh), keep.ctypes.data_as(_lp))
    return keep[:k].copy()


def ref_boxes_iou_bev(a, b):
    """The UNMODIFIED reference TU (oracle/_ref); None if it is not built."""
    r = load_ref()
    return None if r is None else _pair(r.ref_boxes_iou_bev_cpu, a, b)


def ref_boxes_overlap_bev(a, b):
    r = load_ref()
    return None if r is None else _pair(r.ref_boxes_overlap_bev_cpu, a, b)


def random_boxes(n, seed, spread=50.0):
    """SURVEY.md 8d config 1 box generator: xy~U(-spread,spread), z=-1, dx~U(2,6), dy~U(1,2.5), dz=1.5, heading~U(-pi,pi)."""
    r = np.random.default_rng(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = r.uniform(-spread, spread, (n, 2))
    b[:, 2] = -1.0
    b[:, 3] = r.uniform(2, 6, n)
    b[:, 4] = r.uniform(1, 2.5, n)
    b[:, 5] = 1.5
    b[:, 6] = r.uniform(-np.pi, np.pi, n)
    scores = r.uniform(0, 1, n).astype(np.float32)
    return b, scores
