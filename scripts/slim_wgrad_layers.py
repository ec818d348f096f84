"""Every weight-gradient launch of one eager SLIM training step (120k points, 512^2): shape, time (hipEvents), TFLOP/s.
python scripts/slim_wgrad_layers.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.datasets.synthetic import slim_pair  # noqa: E402
from liso_amd.trainer import SlimTrainer  # noqa: E402
from liso_amd.utils import mfma_conv as MC  # noqa: E402
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
torch.manual_seed(0)
tr = SlimTrainer(cfg, dev, use_graph=False)
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
for _ in range(3):
    tr.step(s0, s1)
torch.cuda.synchronize()
log = []
for name in ("conv_wgrad", "conv_wgrad_sparse", "conv_dgrad", "conv_forward"):
    inner = getattr(MC, name)

    def wrapped(*a, _inner=inner, _name=name, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = _inner(*a, **k)
        e1.record()
        x = a[0]
        if _name == "conv_wgrad":
            shape, flop = (tuple(x.shape), tuple(a[2])), 2.0 * x.shape[0] * a[1].shape[2] * a[1].shape[3] * a[2][0] * a[2][1] * a[2][2] * a[2][3]
        elif _name == "conv_wgrad_sparse":
            shape, flop = (tuple(x.shape), tuple(a[3])), 0.0
        elif _name == "conv_dgrad":
            shape, flop = (tuple(x.shape), tuple(a[1].shape)), 2.0 * x.shape[0] * x.shape[2] * x.shape[3] * a[1].numel()
        else:
            w = a[1]
            spec = a[3]
            ho, wo = spec.out_hw(x.shape[2], x.shape[3])
            shape, flop = (tuple(x.shape), tuple(w.shape)), 2.0 * x.shape[0] * ho * wo * w.numel()
        log.append((_name, shape, flop, e0, e1))
        return r

    setattr(MC, name, wrapped)
tr.step(s0, s1)
torch.cuda.synchronize()
agg = {}
for name, shape, flop, e0, e1 in log:
    k = (name, shape)
    a = agg.setdefault(k, [0, 0.0, flop])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
for kind in ("conv_wgrad", "conv_wgrad_sparse", "conv_dgrad", "conv_forward"):
    rows = sorted(((v[1], k, v) for k, v in agg.items() if k[0] == kind), reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"{kind}: {sum(v[0] for _, _, v in rows)} launches, {tot / 1e3:.3f} ms (eager, event-timed incl. host-side gaps)")
    for t, k, v in rows[:14]:
        print(f"   {v[0]:3d} x {t / v[0]:7.1f} us = {t / 1e3:6.3f} ms   x{k[1][0]}  w{k[1][1]}" + (f"   {v[2] / (t / v[0]) / 1e6:6.1f} TFLOP/s" if v[2] else ""))
