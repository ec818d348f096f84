"""What would the pipelined loop gain if a stage cost nothing?  Stage A (SLIM inference) and / or stage B (box mining) replaced by their
cached first results (no GPU work, hardly any host work): upper bounds for optimising either.  python scripts/loop_upper_bounds.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from liso_amd.datasets.synthetic import slim_pair  # noqa: E402
from liso_amd.trainer import LisoLoopTrainer  # noqa: E402
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

dev = torch.device("cuda:0")
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
pairs = [slim_pair(2 + 100 * i, dev, n_points=120000, grid=512, bev_range_m=100.0) for i in range(16)]  # (uniform clouds: one signature)
batch, n_up = 2, 11


def run(fake_a, fake_b):
    torch.manual_seed(0)
    tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=512, use_graph=True, overlap=True, infer_batch=4, flow_ahead=2)
    ctr = [0]

    def steps(n):
        for _ in range(n):
            i = ctr[0] * batch
            ctr[0] += 1
            tr.step_batch([pairs[(i + k) % 16] for k in range(batch)], upcoming=tuple(pairs[(i + k) % 16] for k in range(batch, batch + n_up)))

    steps(12)
    torch.cuda.synchronize()
    if fake_b:
        real_b = tr._mine_from_graph
        cache = {}

        def mine(sample_t0, flow, side):
            if "r" not in cache:
                cache["r"] = real_b(sample_t0, flow, side)
            return cache["r"]

        tr._mine_from_graph = mine
    if fake_a:
        real_a = tr._infer_flow_padded
        cache_a = {}

        def infer(s0, s1):
            key = s0["pcl_ta"]["pcl"].shape
            if key not in cache_a:
                cache_a[key] = real_a(s0, s1).clone()
            return cache_a[key]

        tr._infer_flow_padded = infer
    steps(12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps(40)
    torch.cuda.synchronize()
    print(f"stage A {'cached' if fake_a else 'real  '}  stage B {'cached' if fake_b else 'real  '}: {1e3 * (time.perf_counter() - t0) / 40:.3f} ms per step", flush=True)
    del tr


for fa, fb in ((False, False), (False, True), (True, False), (True, True)):
    run(fa, fb)

# ---- is the detector-only pipeline (both stages cached) bound by the host?  host seconds per step without any synchronisation ----------
torch.manual_seed(0)
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=512, use_graph=True, overlap=True, infer_batch=4, flow_ahead=2)
ctr = [0]


def steps(n):
    for _ in range(n):
        i = ctr[0] * batch
        ctr[0] += 1
        tr.step_batch([pairs[(i + k) % 16] for k in range(batch)], upcoming=tuple(pairs[(i + k) % 16] for k in range(batch, batch + n_up)))


steps(12)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps(40)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"real pipeline: host returns from 40 steps after {1e3 * host / 40:.3f} ms per step, GPU done after {1e3 * (time.perf_counter() - t0) / 40:.3f}", flush=True)
pcls = [c for p_ in pairs[:batch] for c in p_[0]["pcl_full_no_ground_ta"]]
targets = tr._targets_from_flow(pairs[0][0], tr._infer_flow(*pairs[0]))[0]
targets = {k: torch.cat([v, v], dim=0) for k, v in targets.items()}
for _ in range(5):
    tr.detector.step(pcls, targets)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    tr.detector.step(pcls, targets)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"detector.step alone: host {1e3 * host / 40:.3f} ms per step, GPU done after {1e3 * (time.perf_counter() - t0) / 40:.3f}", flush=True)
