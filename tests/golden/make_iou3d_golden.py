"""Generates tests/golden/iou3d_*.npz from the UNMODIFIED reference TU (oracle/_ref/libiou3d_ref.so, built by
oracle/Makefile from /root/reference/iou3d_nms/src/iou3d_cpu.cpp).  Run in the build container only:

    python tests/golden/make_iou3d_golden.py

Fixtures are plain arrays: input boxes, the reference's IoU / overlap matrices, and keep lists obtained by running
the reference's greedy sweep (iou3d_nms.cpp:113-132, restated in oracle/iou3d_oracle.c) over the reference IoU.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import iou3d as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def adversarial():
    b = []
    base = [0.0, 0.0, -1.0, 4.0, 2.0, 1.5, 0.3]
    b.append(base)                                  # identical pair
    b.append(base)
    b.append([4.0, 0.0, -1.0, 4.0, 2.0, 1.5, 0.0])  # touching edge with the next
    b.append([8.0, 0.0, -1.0, 4.0, 2.0, 1.5, 0.0])
    b.append([20.0, 20.0, -1.0, 4.0, 2.0, 1.5, 0.0])            # 90 degree cross
    b.append([20.0, 20.0, -1.0, 4.0, 2.0, 1.5, np.pi / 2])
    b.append([-20.0, 5.0, -1.0, 6.0, 3.0, 1.5, 0.7])            # concentric, nested
    b.append([-20.0, 5.0, -1.0, 2.0, 1.0, 1.5, 0.7])
    b.append([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])               # zeroed "invalid" rows (nms_iou.py:239-241)
    b.append([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    b.append([30.0, -30.0, -1.0, 4.0, 2.0, 1.5, np.pi])         # heading wrap
    b.append([30.0, -30.0, -1.0, 4.0, 2.0, 1.5, -np.pi])
    b.append([30.5, -30.2, -1.0, 4.0, 2.0, 1.5, np.pi / 4])
    b.append([-40.0, -40.0, -1.0, 4.0, 2.0, 1.5, 1e-4])         # tiny relative rotation
    b.append([-40.0, -40.0, -1.0, 4.0, 2.0, 1.5, -1e-4])
    b.append([-40.0, -38.0, -1.0, 4.0, 2.0, 1.5, 0.0])          # shares an edge line exactly
    b.append([10.0, -10.0, -1.0, 4.0, 2.0, 1.5, 0.5])           # corner just touching (within the 1e-2 margin)
    b.append([10.0 + 4.005, -10.0, -1.0, 4.0, 2.0, 1.5, 0.5])
    return np.asarray(b, np.float32)


def emit(name, boxes, scores):
    order = np.argsort(-scores, kind="stable")
    sb = boxes[order]
    iou = O.ref_boxes_iou_bev(sb, sb)
    ov = O.ref_boxes_overlap_bev(sb, sb)
    assert iou is not None, "oracle/_ref not built: run `make -C oracle ref` with /root/reference present"
    keeps = {f"keep_{int(t*100):03d}": O.nms_from_iou(iou, t) for t in (0.1, 0.3, 0.7)}
    np.savez_compressed(os.path.join(OUT, f"iou3d_{name}.npz"), boxes_sorted=sb, iou=iou, overlap=ov, **keeps)
    print(name, sb.shape, "nnz", int((iou > 0).sum()), {k: len(v) for k, v in keeps.items()})


if __name__ == "__main__":
    for n, seed, spread in [(8, 11, 6.0), (256, 1, 50.0), (1000, 3, 50.0), (512, 5, 8.0)]:
        b, s = O.random_boxes(n, seed, spread)
        emit(f"rand{n}_s{seed}", b, s)
    adv = adversarial()
    emit("adversarial", adv, np.linspace(1.0, 0.1, len(adv)).astype(np.float32))
