"""Minimum-jerk track smoothing: the one-launch device optimisation against the numpy restatement of the reference's loop (the
reference itself: 1.8-2.0 s per batch of 3 x 24 frames on the host, tests/golden/make_track_smoothing_golden.py).  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.tracker.track_smoothing import smooth_track_jerk
from oracle import track_smoothing as ot

rs = np.random.default_rng(0)
out = {}
for B, T in ((3, 24), (64, 100), (512, 200)):
    t = np.arange(T)[None, :, None]
    obs = (np.concatenate([0.9 * t + rs.uniform(-30, 30, (B, 1, 1)), 0.2 * t + rs.uniform(-30, 30, (B, 1, 1)), np.zeros((B, T, 1))], -1)
           + rs.normal(0, 0.2, (B, T, 3))).astype(np.float32)
    val = np.ones((B, T), bool)
    yaw = np.zeros((B, T, 1), np.float32)
    args = (torch.from_numpy(obs).cuda(), torch.from_numpy(val).cuda(), torch.from_numpy(yaw).cuda(), 0.1)
    smooth_track_jerk(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        smooth_track_jerk(*args)
    torch.cuda.synchronize()
    dev_ms = 1e3 * (time.perf_counter() - t0) / 5
    cpu_ms = None
    if B <= 64:
        t0 = time.perf_counter()
        ot.smooth_track_jerk(obs, val, yaw, 2000)
        cpu_ms = 1e3 * (time.perf_counter() - t0)
    out[f"B{B}_T{T}"] = {"device_ms_2000_steps": round(dev_ms, 2), "numpy_oracle_ms": None if cpu_ms is None else round(cpu_ms, 1)}
print(json.dumps(out))
