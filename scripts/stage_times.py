"""Stand-alone GPU time of the LISO loop's three pipeline stages (each alone on the GPU, synchronised), per sweep pair:
  python scripts/stage_times.py  -> stage A (SLIM inference replay) at 1 / 2 / 4 / 8 pairs per replay, stage B (box mining graph),
  stage C (detector step, batch 2).  Their sum against the pipelined step of bench.py says how much the three streams overlap."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from liso_amd.datasets.synthetic import slim_pair  # noqa: E402
from liso_amd.trainer import LisoLoopTrainer  # noqa: E402
from liso_amd.utils import mfma_conv as MC  # noqa: E402
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=256, use_graph=True, overlap=True, infer_batch=8, flow_ahead=2)
pairs = [slim_pair(2 + 100 * i, dev, n_points=120000, grid=512, bev_range_m=100.0) for i in range(8)]
shared = len(sys.argv) > 1 and sys.argv[1] == "shared"


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


import contextlib  # noqa: E402

ctx = MC.shared_gpu if shared else contextlib.nullcontext
with ctx():
    for nb in (1, 2, 4, 8):
        def a():
            tr._flows.clear()
            tr._stage_a(*pairs[:nb])
        ms = timed(a)
        print(f"stage A, {nb} pairs per replay: {ms:7.3f} ms = {ms / nb:6.3f} ms per pair", flush=True)
    tr._flows.clear()
    tr._stage_a(*pairs[:2])

    def b():
        tr._mined.clear()
        keep = list(tr._flows)
        tr._stage_b(pairs[0])
        tr._flows[:] = keep
    ms = timed(b)
    print(f"stage B, one pair: {ms:7.3f} ms", flush=True)
    tr._mined.clear()
    for p_ in pairs[:2]:
        tr._stage_b(p_)
    torch.cuda.synchronize()
    got = [tr._take_mined(p_, torch.cuda.current_stream()) for p_ in pairs[:2]]
    from liso_amd.trainer import _BatchedTargets
    targets = _BatchedTargets([g[0] for g in got])
    pcls = [c for p_ in pairs[:2] for c in p_[0]["pcl_full_no_ground_ta"]]
    ms = timed(lambda: tr.detector.step(pcls, targets), n=20)
    print(f"stage C, detector step on 2 clouds: {ms:7.3f} ms", flush=True)
